"""Launch single convs through the C-ABI (fp16) so that a rocprofv3 --kernel-trace of this script gives per-kernel times.
    python tools/conv_time.py n,d,cin,cout,stride [...]"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
amd = importlib.import_module("automated-brain-mri-analysis-and-report-generation-with-retrieval-augmented-clinical-assistance_amd")
rs = np.random.RandomState(0)
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(8, 64, 64, 64, 1)]
for (n, d, cin, cout, st) in cases:
    x = torch.from_numpy(rs.standard_normal((n, d, d, d, cin)).astype(np.float16)).cuda()
    wt = (rs.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(cin * 27)).astype(np.float32)
    b = rs.standard_normal(cout).astype(np.float32)
    for _ in range(6):
        y = amd.ops.conv3d_ndhwc(x, wt, b, stride=st, act=1, slope=0.01)
    torch.cuda.synchronize()
    print("case", n, d, cin, cout, st, "GFLOP", 2.0 * n * (d // st) ** 3 * cin * cout * 27 / 1e9, flush=True)
