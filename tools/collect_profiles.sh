#!/bin/bash
# Copies what tools/profile_round.sh left under gpurun_out/profiles_<tag>/ (merged back by gpurun) into profiles/ under the round's names and
# rebuilds profiles/valu.json, profiles/traffic.json and the kernel resource usage list.  Usage (from the repo root): tools/collect_profiles.sh r5d r5
set -euo pipefail
TAG=${1:?tag of the profile_round.sh run}; R=${2:?round prefix, e.g. r5}
D=gpurun_out/profiles_$TAG
python3 tools/make_valu.py $D 4096 480000 5 profiles/valu.json
python3 tools/make_traffic.py $D 4096 480000 5 profiles/traffic.json
cp $D/trace_summary.md profiles/${R}_one_at_a_time_kernel_trace_stats.md
cp $D/trace_bench_line.json profiles/${R}_one_at_a_time_bench_line_under_rocprof.json
cp $D/trace_default_summary.md profiles/${R}_default_command_kernel_trace_stats.md
cp $D/trace_default_bench_line.json profiles/${R}_default_command_bench_line_under_rocprof.json
cp $D/trace_config2_summary.md profiles/${R}_config2_kernel_trace_stats.md
cp $D/trace_config2_bench_line.json profiles/${R}_config2_bench_line_under_rocprof.json
cp $D/fetch_summary.md profiles/${R}_pmc_fetch_size.md
cp $D/write_summary.md profiles/${R}_pmc_write_size.md
cp $D/sq_summary.md profiles/${R}_pmc_sq.md
cp $D/clk_summary.md profiles/${R}_pmc_clk.md
cp $D/clk_k1_summary.md profiles/${R}_pmc_clk_k1_alone.md
cp $D/clock_probe_default.tsv profiles/${R}_clock_probe_two_batch.tsv
grep -h "timed region" $D/clock_probe_default.err > profiles/${R}_clock_probe_two_batch_timed_region.txt
make -C m17-cxx-demod_amd/csrc asm 2>&1 | grep "remark:" | python3 -c "
import sys, re
out = []
for l in sys.stdin:
    m = re.match(r'remark: [^ ]+ (.*?) \[-Rpass-analysis=kernel-resource-usage\]', l.rstrip())
    if not m: continue
    t = m.group(1)
    if t.startswith('Function Name: '):
        if out: out.append('')
        out.append(t[len('Function Name: '):] + ' [-Rpass-analysis=kernel-resource-usage]')
    elif 'Dynamic Stack' in t: continue
    else: out.append('   ' + t.strip())
print('\n'.join(out))" > profiles/${R}_kernel_resource_usage.txt
ls -la profiles | head -30
