#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c
cd $R && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_r05.py tests/test_gpu_golden.py tests/test_gpu_gemm.py -q -m gpu -k "r05 or production or comm or multigraph or sampler or subgraph or gemm or small_algebra" > $O/tests_a.log 2>&1
tail -8 $O/tests_a.log
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_launch.py tests/test_gpu_model.py -q -m gpu -k "subgraph or batch or csr or launch" > $O/tests_b.log 2>&1
tail -4 $O/tests_b.log
timeout 300 python bench.py --mode minibatch --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_minibatch.json 2> $O/bench_minibatch.err
python -c "
import json; d=json.loads(open('$O/bench_minibatch.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d['minibatch']))"
SGF_SUBGRAPH_CSR=0 timeout 300 python bench.py --mode minibatch --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_minibatch_old.json 2> $O/bench_minibatch_old.err
python -c "
import json; d=json.loads(open('$O/bench_minibatch_old.json').read().strip().splitlines()[-1]); print('old path', d['value'], d['ms_per_step'])"
timeout 300 python -m cProfile -o $O/mb.prof bench.py --mode minibatch --steps 2 --warmup 1 --no-cpu-baseline > $O/mb_prof.log 2>&1
python -c "
import pstats; p=pstats.Stats('$O/mb.prof'); p.sort_stats('tottime').print_stats(45)" > $O/mb_prof_tottime.txt 2>&1
python -c "
import pstats; p=pstats.Stats('$O/mb.prof'); p.sort_stats('cumulative').print_stats(70)" > $O/mb_prof_cum.txt 2>&1
du -sh $O
