set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_tg; rm -rf $O; mkdir -p $O
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  d=$O/$(echo $set | cut -d' ' -f1); timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 scratch/tg_pmc.py > $d.log 2>&1 || { tail -5 $d.log; }
done
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob('gpurun_out/pmc_tg/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_tile_gemm' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(tot): print(f"{k:32s} {tot[k]/max(n[k],1):16.0f}  (n={n[k]})")
PY
