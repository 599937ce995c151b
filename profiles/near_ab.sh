mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests -q -m gpu -x -k "gptq or golden or lookahead or step" 2>&1 | tail -4
for n in 0 256 512 768; do
GQ_NEAR64_MAXN=$n python bench.py --no-whole-model --no-cpu-baseline --steps 3 2>/dev/null | tail -1 > gpurun_out/r3/near64_$n.json
python - $n <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r3/near64_{sys.argv[1]}.json"))
t=d["trailing_update"]
print(sys.argv[1], d["ms_per_step"], t["whole_alone"], t["loop_ms"])
PY
done
