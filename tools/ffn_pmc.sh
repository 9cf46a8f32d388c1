# PMC counters of the fused feed-forward kernel (and of the two igemm launches it replaces) on one MI355X: tools/ffn_ab.py <samples> 1
# under two rocprofv3 --pmc passes.  usage: bash tools/ffn_pmc.sh [out-dir-name] [samples]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ffnpmc}; N=${2:-64}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python $R/tools/ffn_ab.py $N 1 > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/p2 -- python $R/tools/ffn_ab.py $N 1 > $O/p2.log 2>&1; echo "p2 rc=$?"
cd $R
SDV_PMC_FILTER=${3:-} python tools/pmc_summary.py $O/ffn_pmc.csv $(find $O/p1 $O/p2 -name "*counter_collection.csv") | tail -1
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
grep -E "ffn_geglu|igemm_kernel<4, 2, 2, 5, 64, false, 2, (1|0)>" $O/ffn_pmc.csv | cut -d, -f1,2,6,7,8 
