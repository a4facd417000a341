#!/bin/bash
# The driver's N-rank command rehearsed on ONE GPU: N ranks under torch.distributed.run, every rank on device 0 (SSDR_BENCH_DEVICE=0).
#   tools/rehearse_ranks.sh <N> <out.json> [extra bench flags]      (through gpurun; 8 ranks: ~8 x 16 GiB of buffers in the million extra)
N=${1:-8}; OUT=${2:-gpurun_out/rehearse_${N}ranks.json}; shift; shift
mkdir -p "$(dirname "$OUT")"
export SSDR_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
    bench.py --gpus "$N" --steps 20 --warmup 3 "$@" > "$OUT" 2> "${OUT%.json}.err"
echo "rc=$? $(wc -c < "$OUT") bytes"; tail -c 600 "$OUT"; grep -i "FAILED\|fallback\|error" "${OUT%.json}.err" | head -5
