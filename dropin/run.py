#!/usr/bin/env python
"""Runs one of the reference's drivers on the MI355X modules without editing it:

    cd /path/to/KG_Instance_Segmentation
    python /path/to/this/repo/dropin/run.py train.py --data_dir ... --dataset kaggle

`python train.py` puts the script's own directory FIRST on sys.path, ahead of PYTHONPATH, so `import KGnet` would still pick the
reference's KGnet.py next to the script.  This launcher puts dropin/ (the shims) and the repository root in front, keeps the
script's directory after them for the modules that are not replaced (collater, dataset_*, transforms, ...), and executes the
script as __main__.

`preprocessing` (ground-truth maps, called by the reference inside DataLoader WORKER processes, dataset_base.py:94-97) is NOT
replaced by default: forked workers cannot use the GPU.  KG_GPU_GT=1 opts in to the GPU generator (dropin/optin/) and makes every
`torch.utils.data.DataLoader` of the script run with `num_workers=0`, so that ground truth is produced in the training process."""
import os
import runpy
import sys


def main():
    if len(sys.argv) < 2:
        raise SystemExit("usage: run.py <reference script> [args...]")
    here = os.path.dirname(os.path.abspath(__file__))
    script = os.path.abspath(sys.argv[1])
    sys.path[:0] = [here, os.path.dirname(here), os.path.dirname(script)]
    if os.environ.get("KG_GPU_GT") == "1":
        sys.path.insert(0, os.path.join(here, "optin"))
        import torch.utils.data as tud
        _init = tud.DataLoader.__init__

        def init(self, *a, **k):
            if len(a) > 5:          # num_workers passed positionally (dataset, batch_size, shuffle, sampler, batch_sampler, num_workers)
                a = a[:5] + (0,) + a[6:]
            else:
                k["num_workers"] = 0
            _init(self, *a, **k)
        tud.DataLoader.__init__ = init
    sys.argv = [script] + sys.argv[2:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
