#!/bin/bash
# PMC counters of the bf16 GEMM (three contractions), separate passes, kernel-trace only
O=$PWD/gpurun_out/r3c44
mkdir -p $O
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
CMD="python $REPO/tools/gemm_micro.py"
export VS_MICRO_ONLY=bf16
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "gemm_bf16" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $O/pmc_sq -o pmc -f csv -- $CMD > $O/pmc_sq.log 2>&1; echo "sq rc=$?"
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "gemm_bf16" --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM -d $O/pmc_lds -o pmc -f csv -- $CMD > $O/pmc_lds.log 2>&1; echo "lds rc=$?"
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "gemm_bf16" --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_mem -o pmc -f csv -- $CMD > $O/pmc_mem.log 2>&1; echo "mem rc=$?"
python - <<'PY'
import csv, collections, glob, os
O=os.environ.get("O","/root/repo/gpurun_out/r3c44")
for sub in ("pmc_sq","pmc_lds","pmc_mem"):
    fs=glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True)
    if not fs: print(sub,"no csv"); continue
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"]; k=k[k.find("gemm_bf16_kernel"):][:40]
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in per.items():
        print(sub,k,{c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
find $O -type f ! -name "*.csv" ! -name "*.log" -delete
