#!/bin/bash
# Multi-rank schedule of bench.py rehearsed on ONE GPU: gloo process group, every rank on device 0 (RCCL refuses two ranks
# on one device).  Rates are meaningless; what it checks is that the P-rank path runs with the real kernels and that
# every collected / assembled frame equals the eager render bit for bit.
export SRH_BENCH_SINGLE_DEVICE=1 SRH_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for P in 2 4; do
  echo "## P = $P (default slabs)"
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 --master-port $((29600+P)) bench.py --gpus $P --steps 16 --warmup 4 2> gpurun_out/mr_$P.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('n_gpus','steps','ms_per_step')}, d['config'].get('rows_per_rank'), '|', d['config'].get('collection'), '|', d['config'].get('check')); print('literal_root0' in d, 'links_measured' in d)"
  grep -h "check ok\|FAILED\|differs\|Error\|error" gpurun_out/mr_$P.err | head -5
done
echo "## P = 2 equal slabs, batched views call"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 16 --warmup 4 --slabs contiguous 2> gpurun_out/mr_2c.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('n_gpus','steps','ms_per_step')}, d['config'].get('rows_per_rank'), '|', d['config'].get('collection'), '|', d['config'].get('check'))"
grep -h "check ok\|FAILED\|differs" gpurun_out/mr_2c.err | head -3
echo "## P = 4, work-balanced slabs + one srh_render_views call per batch (the default at 8 ranks)"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 4 --steps 16 --warmup 4 --slabs cost 2> gpurun_out/mr_4c.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('n_gpus','steps','ms_per_step')}, d['config'].get('rows_per_rank'), '|', d['config'].get('launches'), '|', d['config'].get('collection'), '|', d['config'].get('check'))"
grep -h "check ok\|FAILED\|differs" gpurun_out/mr_4c.err | head -3
echo "## P = 2, a check failure injected on rank 1: the headline must come from the literal gather-to-rank-0 leg"
SRH_BENCH_LITERAL=1 SRH_BENCH_INJECT_CHECK_FAIL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 16 --warmup 4 > gpurun_out/mr_2f.out 2> gpurun_out/mr_2f.err; echo "exit code $? (3 = a schedule failed its check; the line is still printed)"
tail -1 gpurun_out/mr_2f.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('n_gpus','steps','ms_per_step')}, d['config'].get('rows_per_rank'), '|', d['config'].get('collection'), '|', d['config'].get('check')); print('check_failed:', d.get('check_failed'), 'other_schedule:', d.get('other_schedule'))"
grep -h "check ok\|FAILED\|differs\|Error" gpurun_out/mr_2f.err | head -5
