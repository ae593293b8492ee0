set -uo pipefail
cd $GRAFT_REPO_ROOT
L=nmrfit_amd/lib
LIBS="$L/libnmrfit_amd.so $L/libab_r1.so $L/libab_r2.so $L/libab_r3.so $L/libab_r4.so $L/libab_r5.so $L/libab_r6.so $L/libab_r7.so"
python tools/ab.py $LIBS --variant 6 > gpurun_out/ablate_time.txt 2>&1
export TMPDIR=/tmp
rm -rf /tmp/abp; 
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/abp -- python3 $GRAFT_REPO_ROOT/tools/ab_pmc.py --variant 6 $(for l in $LIBS; do echo $GRAFT_REPO_ROOT/$l; done) > $GRAFT_REPO_ROOT/gpurun_out/ablate_pmc.log 2>&1)
python3 tools/ab_pmc.py --summarise /tmp/abp $LIBS > gpurun_out/ablate_pmc.txt 2>&1
cat gpurun_out/ablate_time.txt gpurun_out/ablate_pmc.txt
