"""CPU oracle for the BraTS nnU-Net sliding-window predictor.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
directory; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the timed CPU baseline.

What is pinned and what is not ("Oracle"):

* ``unet_ref``  - restatement of ``model_architecture/generic_UNet.py`` in plain
  torch-CPU fp32 functional ops.  PINNED: checked against the reference module
  itself (imported from /root/reference through ``ref_shim``) and against the
  golden vectors that ``gen_golden.py`` produced from that import
  (``tests/golden/net_*.npz``).
* ``tiler_ref`` - restatement of the *un-vendored* nnU-Net v1 runtime the
  reference driver calls (sliding-window steps, Gaussian importance map,
  mirror TTA, aggregation, region export, preprocessing).  The source of that
  package is absent from /root/reference, the reference holds no tests or
  fixtures for it: PARITY UNPINNED for these functions (cross-checked only
  against scipy and the tile tables implied by ``data/temp_inference_output1``).
* ``driver_ref`` - restatement of the parts of
  ``run_brats2021_inference_singlethread.py`` that are pure numpy (fold mean,
  label-round ensemble, volume print).  Truth tables are derived by running the
  reference's own numpy expression (``np.round((a+b)/2)``), so they are pinned
  by construction.
"""
