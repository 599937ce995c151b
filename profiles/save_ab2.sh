mkdir -p gpurun_out/r3
for mode in ${MODES:-"GQ_X=0" "GQ_SAVER_POLL=1" "GQ_BLOCK_SYNC=1" "GQ_SAVE_SKIP=1"}; do
  env $mode python bench.py --workload llama3-8b-model-q4k --steps 1 --warmup 0 --fused-forward 2>/dev/null | tail -1 > gpurun_out/r3/save_ab.json
  python - "$mode" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r3/save_ab.json")); w=d.get("whole_model",d)
print(sys.argv[1], w["wall_s_quantizer_region"], "gpu", {k:round(v,2) for k,v in w["split"]["gpu_s"].items()}, "host", {k:round(v,2) for k,v in w["split"]["host_s"].items()}, w["split"].get("saver_copy_thread_busy_s"), w["split"].get("saver_writer_process_busy_s"), w["split"].get("allocator"), w["split"].get("kernel_ms_launches"))
PY
done
