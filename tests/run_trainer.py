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


def main():
    mode, trainer, args = sys.argv[1], os.path.abspath(sys.argv[2]), sys.argv[3:]
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
