set -x
mkdir -p gpurun_out/r06
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-whole-model --no-side-legs 2>gpurun_out/r06/base_bench.err | tail -1 > gpurun_out/r06/base_bench_compact.json
cat gpurun_out/r06/base_bench_compact.json
GQ_PROF_DUMP=gpurun_out/r06/tl_base.txt python bench.py --steps 3 --warmup 2 --breakdown --no-cpu-baseline --no-whole-model --no-side-legs 2>gpurun_out/r06/base_breakdown.err | tail -1
python profiles/timeline.py gpurun_out/r06/tl_base.txt > gpurun_out/r06/timeline_base.txt 2>&1; head -60 gpurun_out/r06/timeline_base.txt
bash profiles/trace_chain.sh > gpurun_out/r06/trace_chain_base.txt 2>&1; cat gpurun_out/r06/trace_chain_base.txt
bash profiles/r06_syrk_pmc.sh
