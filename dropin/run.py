#!/usr/bin/env python
"""Runs one of the reference's drivers on the MI355X modules without editing it:

    cd /path/to/KG_Instance_Segmentation
    python /path/to/this/repo/dropin/run.py train.py --data_dir ... --dataset kaggle

`python train.py` puts the script's own directory FIRST on sys.path, ahead of PYTHONPATH, so `import KGnet` would still pick the
reference's KGnet.py next to the script.  This launcher puts dropin/ (the shims) and the repository root in front, keeps the
script's directory after them for the modules that are not replaced (collater, dataset_*, transforms, ...), and executes the
script as __main__."""
import os
import runpy
import sys


def main():
    if len(sys.argv) < 2:
        raise SystemExit("usage: run.py <reference script> [args...]")
    here = os.path.dirname(os.path.abspath(__file__))
    script = os.path.abspath(sys.argv[1])
    sys.path[:0] = [here, os.path.dirname(here), os.path.dirname(script)]
    sys.argv = [script] + sys.argv[2:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
