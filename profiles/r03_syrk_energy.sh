#!/bin/bash
# a counter-backed account of where the SYRK's power goes.  For the shipped kernel and its variants
# (separate rocprofv3 --pmc passes, --kernel-trace only) on the widest bench input (C = 14336, 65536 tokens):
#   clock (GRBM_GUI_ACTIVE / duration), MFMA busy, LDS instructions, L2 requests / hits / misses, L2-miss bytes (FETCH_SIZE)
# and, without the profiler, the board power and sclk sampled by rocm-smi while the same launch repeats for ~3 s --
# next to the register-resident MFMA loop (profiles/micro/mfma_power.hip: the power ceiling of the instruction itself).
# usage (GPU box): bash profiles/r03_syrk_energy.sh   -> gpurun_out/r03/r03_syrk_energy.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03; mkdir -p $OUT
LOG=$OUT/r03_syrk_energy.txt; : > $LOG
cd /tmp && export TMPDIR=/tmp
sample_power() {  # $1 = label, rest = command; samples rocm-smi every ~100 ms while the command runs
  local label=$1; shift
  ( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.1; done ) > $OUT/smi_$label.jsonl &
  local pid=$!
  "$@" > $OUT/run_$label.log 2>&1
  kill $pid 2>/dev/null; wait $pid 2>/dev/null
  python3 - <<PY >> $LOG
import json, re
pw, sclk = [], []
for line in open("$OUT/smi_$label.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    for card, v in d.items():
        for k, x in v.items():
            if "Power" in k and "W" in k:
                try: pw.append(float(x))
                except Exception: pass
            if k.startswith("sclk clock speed"):
                m = re.search(r"(\d+)Mhz", str(x))
                if m: sclk.append(int(m.group(1)))
pw.sort(); sclk.sort()
top = pw[len(pw) // 2:] if pw else []
print(f"[$label] rocm-smi: {len(pw)} samples, power median {pw[len(pw)//2] if pw else None} W, upper-half mean {sum(top)/max(len(top),1):.0f} W, max {pw[-1] if pw else None} W; sclk median {sclk[len(sclk)//2] if sclk else None} MHz")
print("[$label] " + " | ".join(l.strip() for l in open("$OUT/run_$label.log") if "TFLOP" in l or "PFLOP" in l)[-400:])
PY
}
pmc_pass() {  # $1 = label, $2 = counters, rest = env assignments
  local label=$1 ctr=$2; shift 2
  rm -rf $OUT/pmc_$label
  env "$@" CS=14336 NSEQ=32 ITERS=2 timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$label -o p -- python $R/profiles/syrk_probe.py > $OUT/pmc_$label.log 2>&1 || echo "[$label] pass failed ($ctr)" >> $LOG
  python3 - <<PY >> $LOG
import csv, glob, collections
agg = collections.defaultdict(float); dur = 0.0; n = 0
for f in glob.glob("$OUT/pmc_$label/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; n += 1
for f in glob.glob("$OUT/pmc_$label/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "syrk16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print(f"[$label] {n} SYRK launches {dur:.2f} ms: " + "  ".join(f"{k}={v:.5g}" for k, v in sorted(agg.items())))
if "GRBM_GUI_ACTIVE" in agg and dur > 0:
    cyc = agg["GRBM_GUI_ACTIVE"] / 8
    extra = f", MFMA busy {agg['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024) * 100:.1f} %" if "SQ_VALU_MFMA_BUSY_CYCLES" in agg else ""
    print(f"[$label]   clock {cyc / dur / 1e6:.3f} GHz{extra}")
if "FETCH_SIZE" in agg: print(f"[$label]   L2-miss reads {agg['FETCH_SIZE'] * 2048 / 1e9 / max(n, 1):.1f} GB per launch (x2 KB gfx950 correction)")
if "TCC_HIT_sum" in agg: print(f"[$label]   L2 hit rate {agg['TCC_HIT_sum'] / max(agg['TCC_HIT_sum'] + agg['TCC_MISS_sum'], 1) * 100:.1f} %")
PY
  rm -rf $OUT/pmc_$label
}
for V in "default:GQ_X=1" "image:GQ_SYRK_IMAGE=1" "nopersist:GQ_SYRK_PERSIST=0"; do
  L=${V%%:*}; E=${V#*:}
  pmc_pass ${L}_clk "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16" $E
  pmc_pass ${L}_l2 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum FETCH_SIZE" $E
  pmc_pass ${L}_lds "SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" $E
  sample_power $L env $E CS=14336 NSEQ=32 ITERS=60 python $R/profiles/syrk_probe.py
done
# the instruction's own power ceiling: register-resident MFMA loops on random data
(cd $R/profiles/micro && [ -x mfma_power ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 mfma_power.hip -o mfma_power) >/dev/null 2>&1
sample_power mfma_loop $R/profiles/micro/mfma_power
cat $LOG
