"""Run one of the reference's trainer scripts UNCHANGED, on CPU, against the stand-ins in
tests/standins (test infrastructure; used by tests/test_trainer_unchanged.py).

    python tests/run_trainer.py reference /root/reference/large/main.py <trainer args>
        -> the reference's own ours.py
    python tests/run_trainer.py ours /root/reference/large/main.py <trainer args>
        -> sgformer_amd.launch (drop-in registered as sys.modules['ours']) with the CPU kernel table
           of tests/cpu_kernels.py installed, since this container has no GPU.  On a GPU box the
           same launcher runs the HIP kernels (`python -m sgformer_amd.launch ...`).
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(HERE, "standins"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def _cuda_as_cpu():
    """100M/nb-sample.py:115 hard-codes `torch.device(f"cuda:{args.device}")` (no --cpu switch).  This container has no
    GPU: SGF_CUDA_AS_CPU=1 makes `torch.device("cuda:N")` in the TRAINER's code name the CPU, in both modes, so that the
    unchanged file can run.  (Only the Python-level name is replaced; tensors and modules see real torch.device objects.)"""
    import torch
    real = torch.device

    class _Device:
        def __new__(cls, *a, **k):
            if a and isinstance(a[0], str) and a[0].startswith("cuda"):
                return real("cpu")
            return real(*a, **k)

    torch.device = _Device


def _fix_100m_import(trainer):
    """100M/nb-sample.py:12 imports `load_fixed_splits` from 100M/data_utils.py, which does not define it: the file cannot
    be imported as shipped.  Pre-import the trainer's own data_utils and add the missing name (a stub: the papers100M path,
    nb-sample.py:96-97, never calls it) — in BOTH modes, so that the comparison runs the same file."""
    tdir = os.path.dirname(trainer)
    if os.path.basename(tdir) != "100M":
        return
    sys.path.insert(0, tdir)
    import data_utils
    if not hasattr(data_utils, "load_fixed_splits"):
        def load_fixed_splits(*a, **k):
            raise NotImplementedError("100M/data_utils.py has no load_fixed_splits")
        data_utils.load_fixed_splits = load_fixed_splits


def main():
    mode, trainer, args = sys.argv[1], os.path.abspath(sys.argv[2]), sys.argv[3:]
    if os.environ.get("SGF_CUDA_AS_CPU") == "1":
        _cuda_as_cpu()
    if mode == "reference":
        _fix_100m_import(trainer)
    if mode == "ours":
        from sgformer_amd import launch, ops
        from tests.cpu_kernels import CpuKernels
        ops.set_kernels(CpuKernels())
        launch.main([trainer] + args)
        print('SGF_OURS_MODULE', sys.modules['ours'].__name__, 'kernels', ops.K.name)
    elif mode == "reference":
        sys.argv = [trainer] + args
        sys.path.insert(0, os.path.dirname(trainer))
        runpy.run_path(trainer, run_name="__main__")
        print('SGF_OURS_MODULE', sys.modules['ours'].__file__)
    else:
        raise SystemExit(f"unknown mode {mode}")


if __name__ == "__main__":
    main()
