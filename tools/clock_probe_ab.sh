probe() {
python bench.py --workload full --steps 4000 --warmup 2 --spinup 0 --no-cpu-baseline --no-extra --no-parity-probe > /tmp/pb.json 2>/dev/null &
BP=$!
sleep 3
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.5; done
wait $BP
python -c "
import sys,json
d=json.loads(open('/tmp/pb.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],4))"
}
echo "== old kernel"; SSDR_FUSED64=0 probe
echo "== new w4"; SSDR_LIB_PATH=$PWD/supersdr_amd/libssdr_w4.so probe
echo "== new w6"; probe
echo "== new w6 no swaps"; SSDR_LIB_PATH=$PWD/supersdr_amd/libssdr_a1.so probe
