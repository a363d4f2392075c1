export GL_DEV_SWITCHES=1 TMPDIR=/tmp
mkdir -p gpurun_out/pmc_wide
K=$PWD/gligen_amd/build/kbench
S=$PWD/tools/unet_b8.shapes
OUT=$PWD/gpurun_out/pmc_wide
run() { local name=$1; local filt=$2; shift 2
  ( cd /tmp && timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- $K $S 2 "$filt" ) > $OUT/$name.log 2>&1
}
for spec in "w32:gemm 8192 5120 640 1" "w16:gemm 2048 10240 1280 1" "h64:conv 8 64 64 320 0 320"; do
  n=${spec%%:*}; f=${spec#*:}
  run ${n}_a "$f" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
  run ${n}_b "$f" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC
  run ${n}_c "$f" SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_LEVEL_LDS SQ_INSTS_SMEM
done
for d in $OUT/*_a $OUT/*_b $OUT/*_c; do f=$(find $d -name "*counter_collection.csv" | head -1); [ -z "$f" ] && continue; python3 - $f $(basename $d) <<PY
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda: collections.defaultdict(float))
disp=collections.defaultdict(set)
for r in rows:
    k=r["Kernel_Name"][:48]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k,v in acc.items():
    if "gemm_wide" in k or "conv_halo" in k:
        print(sys.argv[2], k, len(disp[k]), {c: round(x/len(disp[k])) for c,x in v.items()})
PY
done
