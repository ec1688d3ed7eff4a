#!/bin/bash
# round 4: SQ counters of the dominant bf16x3 layer (M=1,605,632 256->256 3x3), 8-wave tile (TT_X3_PIPE=0) vs hand-pipelined tiles
# (TT_X3_PIPE=1: 2 x 2 waves of 128 x 128, =2: 4 x 1 waves of 64 x 256).  Separate --pmc passes, --kernel-trace only.
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export TT_GLDS_X3_TILE=256
ARMS=${ARMS:-"0 2"}
ACT=${ACT:-0}
PA="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS"
PB="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
PC="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_BUSY_CYCLES"
for arm in $ARMS; do
  for pass in A B C; do
    eval P=\$P$pass
    TT_X3_PIPE=$arm TT_MB_ACT=$ACT timeout 300 rocprofv3 --kernel-trace --pmc $P -d "$OUT/r04_sq_arm${arm}_$pass" -o p --output-format csv -- \
        python $ROOT/tools/conv_microbench.py 64 112 224 256 256 3 1 x3 10 > "$OUT/r04_sq_arm${arm}_$pass.log" 2>&1
  done
done
python - <<PY > $OUT/r04_conv_sq_counters_act$ACT.txt
import csv, glob, os
out = "$OUT"
for arm in "$ARMS".split():
    tot = {}
    n = {}
    dur = []
    for ps in "ABC":
        for f in glob.glob(os.path.join(out, f"r04_sq_arm{arm}_{ps}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "conv_x3_pipe" not in k and "conv_igemm_glds" not in k:
                    continue
                c = r["Counter_Name"]
                tot[c] = tot.get(c, 0.0) + float(r["Counter_Value"])
                n[c] = n.get(c, 0) + 1
                name = k
        for f in glob.glob(os.path.join(out, f"r04_sq_arm{arm}_{ps}", "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "conv_x3_pipe" in r["Kernel_Name"] or "conv_igemm_glds" in r["Kernel_Name"]:
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print(f"== TT_X3_PIPE={arm}: {name[:90]}")
    print(f"   dispatches per pass {max(n.values())}, kernel time under the counters: median {sorted(dur)[len(dur)//2]:.3f} ms")
    avg = {c: tot[c] / n[c] for c in tot}
    for c in sorted(avg):
        print(f"   {c:28s} {avg[c]:16.0f}")
    g = avg.get
    mf = g("SQ_INSTS_MFMA", 0)
    if mf:
        print(f"   derived: MFMA busy cycles / MFMA = {g('SQ_VALU_MFMA_BUSY_CYCLES', 0) / mf:.1f}; MFMA pipe utilisation = busy / (4 x SQ_BUSY_CU_CYCLES) = "
              f"{g('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * g('SQ_BUSY_CU_CYCLES', 1)):.3f}")
        print(f"            non-MFMA VALU per MFMA = {(g('SQ_INSTS_VALU', 0) - mf) / mf:.2f}; SALU per MFMA = {g('SQ_INSTS_SALU', 0) / mf:.2f}; "
              f"LDS instr per MFMA = {g('SQ_INSTS_LDS', 0) / mf:.2f}; VMEM per MFMA = {g('SQ_INSTS_VMEM_RD', 0) / mf:.3f}")
        wc = g("SQ_WAVE_CYCLES", 1)
        print(f"            wave cycles: waiting (s_waitcnt / barrier) {g('SQ_WAIT_ANY', 0) / wc:.3f}, issue-stalled {g('SQ_WAIT_INST_ANY', 0) / wc:.3f} "
              f"(LDS part {g('SQ_WAIT_INST_LDS', 0) / wc:.3f}), issuing {g('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}")
        print(f"            LDS bank-conflict cycles / LDS active cycles = {g('SQ_LDS_BANK_CONFLICT', 0) / max(g('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}")
        print(f"            GRBM_GUI_ACTIVE per dispatch = {g('GRBM_GUI_ACTIVE', 0):.0f} cycles")
PY
cat $OUT/r04_conv_sq_counters_act$ACT.txt
find "$OUT" -path "*r04_sq_arm*" -name "*.csv" -delete
