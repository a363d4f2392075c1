#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call B: where does attn2_kernel's time go? Ablations (GL_ATTN_DBG: 1 no DMA in the loop, 2 no barrier, 3 both, 4 exp2 -> mul,
# 8 no MFMAs), ring depth / block size variants (GL_ATTN_V2 = 1: 4 waves x 2 slots, 2: 8 x 2, 3: 4 x 3, 4: 8 x 3), PMC passes
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r3b
mkdir -p $O
K=$R/gligen_amd/build/kbench
{
for v in 1 2 3 4; do
  echo "== GL_ATTN_V2=$v"; GL_ATTN_V2=$v timeout 120 $K tools/attn.shapes 20 2>&1 | grep "^attn 8 8 40"
done
for d in 1 2 3 4 8; do
  echo "== GL_ATTN_DBG=$d"; GL_ATTN_DBG=$d timeout 120 $K tools/attn1.shapes 20 2>&1 | grep "^attn 8 8 40"
done
} > $O/attn_variants.txt 2>&1
cat $O/attn_variants.txt
run() { local name=$1; shift
  ( cd /tmp && timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$name -- $K $R/tools/attn1.shapes 2 ) > $O/pmc_$name.log 2>&1
}
for v in 1 3; do
  export GL_ATTN_V2=$v
  run v${v}_a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
  run v${v}_b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC
  run v${v}_c SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SMEM
done
O=$O python - <<'PY'
import csv, glob, collections, os
O = os.environ["O"]
for f in sorted(glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/pmc_")[1].split("/")[0])
    for k, v in sorted(agg.items()):
        print("    %-28s %14.0f  (%d dispatches)" % (k, sum(v) / len(v), len(v)))
PY
