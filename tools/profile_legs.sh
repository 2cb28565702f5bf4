#!/bin/bash
# rocprofv3 passes of bench.py's headline and legs (run on the GPU box from the repo root):
#   tools/profile_legs.sh <leg: headline | any name of bench.py LEGS> <kernel regex> <out json> [<kernel regex 2> <out json 2>] [skip]
# <out json> is written straight into profiles/ (so that the bench run below reads it): the PMC summary per (regex, out) pair,
# the --stats csv, and LAST the leg's own unabridged record from a plain run of the same command — whose roofline.traffic is
# therefore the hbm_bytes_per_launch of the summary beside it (one commit, one run: VERDICT r03 item 2).
# The --pmc passes run with --kernel-trace only (never with --stats or another trace domain), each counter group in its own
# pass, as the guide prescribes.
set -o pipefail
leg=$1; rx=$2; out=$3; rx2=$4; out2=$5; skip=${6:-0}
root=$PWD
d=$root/gpurun_out/prof_r04/$leg
mkdir -p $d $(dirname $out)
cd /tmp && export TMPDIR=/tmp && cd $root
if [ "$leg" = headline ]; then cmd="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-legs"; else cmd="bench.py --leg $leg --no-cpu-baseline"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $d/stats -o p -- python3 $cmd > $d/stats.log 2>&1 || { tail -5 $d/stats.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $d/pmc1 -o p -- python3 $cmd > $d/pmc1.log 2>&1 || { tail -5 $d/pmc1.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $d/pmc2 -o p -- python3 $cmd > $d/pmc2.log 2>&1 || { tail -5 $d/pmc2.log; exit 1; }
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SMEM --kernel-trace --output-format csv -d $d/pmc3 -o p -- python3 $cmd > $d/pmc3.log 2>&1 || { tail -5 $d/pmc3.log; exit 1; }
python3 tools/pmc_summary.py $d "$rx" $out "$leg: python3 $cmd" $skip
if [ -n "$rx2" ]; then python3 tools/pmc_summary.py $d "$rx2" $out2 "$leg: python3 $cmd" $skip; fi
cp $(find $d/stats -name "*kernel_stats.csv" | head -1) ${out%_pmc.json}_kernel_stats.csv
# the record itself, from a run without the profiler, after the summaries exist
if [ "$leg" = headline ]; then
  python3 $cmd --full-out ${out%_pmc.json}_bench.json > $d/plain.log 2>&1 || { tail -5 $d/plain.log; exit 1; }
else
  python3 $cmd 2> $d/plain.err | grep '^{' | tail -1 > ${out%_pmc.json}_bench.json
fi
# the raw traces are large (gpurun merges at most 64 MiB back): keep the summaries only
rm -rf $d/pmc1 $d/pmc2 $d/pmc3 $d/stats
