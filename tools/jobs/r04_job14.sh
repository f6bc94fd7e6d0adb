set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
echo "# rocprofv3 --pmc pass over tools/prof_ops.py --run --batch 4 (512x512, fp32), joined with the launch plan by tools/pmc_ops.py <dir> 4" > $O/r04_pmc_waves_b4.txt
echo "# SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (us * 2400 cycles) = fraction of the time a SIMD's matrix pipe is busy" >> $O/r04_pmc_waves_b4.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_b4 -- python3 $R/tools/prof_ops.py --run --batch 4 > $O/r04_pmc_b4.log 2>&1
python3 $R/tools/pmc_ops.py $O/pmc_b4 4 >> $O/r04_pmc_waves_b4.txt
rm -rf $O/pmc_b4
python3 - <<PY
import re
tot_us = 0; busy = 0; mm_us = 0; mm_busy = 0
for l in open("$O/r04_pmc_waves_b4.txt"):
    if l.startswith("#") or l.startswith("op "): continue
    f = l.split()
    if len(f) < 6: continue
    us = float(f[1]); b = float(f[4])
    tot_us += us; busy += b
    if b > 0: mm_us += us; mm_busy += b
print("batch 4: traced frame %.0f us; matrix-pipe busy %.1f %% time-weighted over the whole step, %.1f %% over the launches that use it" % (tot_us, 100 * busy / 1024 / (tot_us * 2400), 100 * mm_busy / 1024 / (mm_us * 2400)))
PY
