#!/usr/bin/env python
"""Drop-in for the reference's ``run_brats2021_inference_singlethread.py`` (same file name, same
``--input <dir> --output <dir>`` contract, same products and exit codes), so that
``run_full_pipeline.py:162-168`` runs the MI355X-native path without an edit.  The models are
looked up under ``<this dir>/nnUNet_results`` exactly as the reference does (:253-264)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    import brats_amd
    sys.exit(brats_amd.driver.main(script_dir=os.path.dirname(os.path.abspath(__file__))))
