#!/bin/bash
# Round profiles of the bench solve on a GPU box: rocprofv3 kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate
# passes), SQ counters (one pass each), batch sweep, per-configuration timings.  usage: tools/profile_round.sh <tag>
tag=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python $R/bench.py --steps 50 --no-cpu-baseline --no-extra > $O/bench_prof.json 2> $O/bench_prof.err
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/${tag}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o run -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
done
python - <<PY > $O/${tag}_pmc_counters.txt
import csv, glob
print('PMC counters of the K2 solve kernel (one rocprofv3 --pmc pass per counter, python bench.py --steps 5 --warmup 2), mean per dispatch:')
for d in sorted(glob.glob('$O/pmc_*')):
    f = glob.glob(d + '/*counter_collection.csv')
    if not f: continue
    vals = {}
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if 'snsde_m4_kernel' in k or 'snsde_mfma_kernel' in k:
            vals.setdefault((k[:70], r['Counter_Name']), []).append(float(r['Counter_Value']))
    for (k, c), v in vals.items():
        print(f'{c:28s} {sum(v)/len(v):16.1f}   n={len(v):3d}   {k}')
PY
python $R/tools/sweep_batch.py > $O/${tag}_batch_sweep.txt 2>&1
python $R/tools/time_configs.py > $O/${tag}_time_configs.txt 2>&1
python $R/tools/time_train.py > $O/${tag}_time_train.txt 2>&1
python $R/tools/time_k5.py > $O/${tag}_time_h256.txt 2>&1
python $R/tools/time_recompute.py > $O/${tag}_recompute_modes.txt 2>&1
python $R/tools/time_wrapper.py > $O/${tag}_time_wrapper.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -o run -- python $R/tools/time_configs.py > /dev/null 2>&1
cp $(find $O/stats_train -name '*kernel_stats.csv' | head -1) $O/${tag}_train_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wrap -o run -- python $R/tools/prof_wrapper.py 30 > /dev/null 2>&1
cp $(find $O/stats_wrap -name '*kernel_stats.csv' | head -1) $O/${tag}_wrapper_step_kernel_stats.csv
cat $O/bench_prof.json | head -c 600; echo; cat $O/${tag}_pmc_counters.txt; cat $O/${tag}_batch_sweep.txt; tail -15 $O/${tag}_time_configs.txt; tail -12 $O/${tag}_time_train.txt
