#!/usr/bin/env python
"""Drop-in for the reference's ``run_brats2021_inference_singlethread.py`` (same file name, same
``--input <dir> --output <dir>`` contract, same products and exit codes), so that
``run_full_pipeline.py:162-168`` runs the MI355X-native path without an edit.  The models are
looked up under ``<this dir>/nnUNet_results`` exactly as the reference does (:253-264).

When a resident worker (``python -m brats_amd.worker``) listens on the default socket the request is handed to it -
interpreter start-up, ``import torch``, the library and the checkpoints are then already paid for - and this process only
relays its output and return code; otherwise the path runs in this process (MI355_NO_WORKER=1 forces that)."""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(HERE, "automated-brain-mri-analysis-and-report-generation-with-retrieval-augmented-clinical-assistance_amd")


def _worker_client():
    """worker.py loaded by path: importing the package would import torch, which is what the worker saves."""
    spec = importlib.util.spec_from_file_location("_mi355_worker_client", os.path.join(PKG, "worker.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    argv = sys.argv[1:]
    if os.environ.get("MI355_NO_WORKER") != "1":
        client = _worker_client()
        if os.path.exists(client.default_socket_path()) and not any(a in ("-h", "--help") for a in argv):
            rc = client.request(client.absolutise(argv), HERE)
            if rc is not None:
                sys.exit(rc)
    sys.path.insert(0, HERE)
    import brats_amd
    sys.exit(brats_amd.driver.main(script_dir=HERE))
