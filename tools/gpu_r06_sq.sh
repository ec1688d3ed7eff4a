#!/bin/bash
# round 6: SQ counters (MFMA pipe utilisation) of the dominant 3 x 3 layer M=1,605,632 256->256 on the three kernels that run it:
# bf16x3 run-staged (x3), the same with pre-split activations (x3p), the h2 kernel.  Separate --pmc passes, --kernel-trace only.
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PA="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS"
PB="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for mode in x3 x3p h2; do
  for pass in A B; do
    eval P=\$P$pass
    timeout 300 rocprofv3 --kernel-trace --pmc $P -d "$OUT/r06_sq_${mode}_$pass" -o p --output-format csv -- \
        python $ROOT/tools/conv_microbench.py 64 112 224 256 256 3 1 $mode 10 > "$OUT/r06_sq_${mode}_$pass.log" 2>&1
  done
done
cd $ROOT
python - <<PY > $OUT/r06_conv_sq_counters.txt
import csv, glob, os
out = "$OUT"
print("Round 6: SQ counters of the 3 x 3 layer M=1,605,632 256->256 (tools/conv_microbench.py 64 112 224 256 256 3 1 <mode> 10), two --pmc passes each")
for mode in ("x3", "x3p", "h2"):
    tot, n, dur, name = {}, {}, [], "?"
    for ps in "AB":
        for f in glob.glob(os.path.join(out, f"r06_sq_{mode}_{ps}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "conv_x3" not in k and "conv_h2" not in k and "conv_igemm_glds" not in k:
                    continue
                c = r["Counter_Name"]
                tot[c] = tot.get(c, 0.0) + float(r["Counter_Value"])
                n[c] = n.get(c, 0) + 1
                name = k
        for f in glob.glob(os.path.join(out, f"r06_sq_{mode}_{ps}", "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if any(t in r["Kernel_Name"] for t in ("conv_x3", "conv_h2", "conv_igemm_glds")):
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    dur.sort()
    print(f"== {mode}: {name[:100]}")
    if not dur:
        print("   no data"); continue
    print(f"   kernel time under the counters: median {dur[len(dur) // 2]:.3f} ms")
    for c in sorted(tot):
        print(f"   {c:32s} {tot[c] / n[c]:14.0f}   (per dispatch)")
    g = lambda c: tot.get(c, 0.0) / max(n.get(c, 1), 1)
    if g("SQ_INSTS_MFMA") and g("SQ_BUSY_CU_CYCLES"):
        print(f"   derived: MFMA busy cycles / MFMA = {g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_INSTS_MFMA'):.1f}; MFMA pipe utilisation = busy / (4 x SQ_BUSY_CU_CYCLES) = {g('SQ_VALU_MFMA_BUSY_CYCLES') / (4 * g('SQ_BUSY_CU_CYCLES')):.3f}")
        print(f"            non-MFMA VALU per MFMA = {(g('SQ_INSTS_VALU') - g('SQ_INSTS_MFMA')) / g('SQ_INSTS_MFMA'):.2f}; LDS instr per MFMA = {g('SQ_INSTS_LDS') / g('SQ_INSTS_MFMA'):.2f}")
PY
cat $OUT/r06_conv_sq_counters.txt | cut -c1-170
rm -rf $OUT/r06_sq_*
