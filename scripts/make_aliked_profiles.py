"""profiles/<round>_aliked_* from the rocprofv3 outputs of scripts/gpu_aliked_profile.sh <tag> pmc (merged into gpurun_out/):
kernel stats (copied), HBM bytes per kernel (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE) and the MFMA / clock summary —
ALIKED at the config-5 tile size (8 tiles of 1500 x 1000 per launch sequence).

    python scripts/make_aliked_profiles.py r03 r03f
"""
import collections, csv, json, re, shutil, sys
rnd, tag = sys.argv[1], sys.argv[2]
def clean(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); return n.split('(')[0].replace(', ', ',')
shutil.copy(f'gpurun_out/prof_aliked_{tag}/aliked_kernel_stats.csv', f'profiles/{rnd}_aliked_kernel_stats.csv')
rows = list(csv.DictReader(open(f'gpurun_out/prof_aliked_{tag}/aliked_kernel_stats.csv')))
total_ms = sum(float(r['TotalDurationNs']) for r in rows) / 1e6
out = {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.defaultdict(lambda: [0, 0.0])
    try:
        for r in csv.DictReader(open(f'gpurun_out/pmc_aliked_{tag}_{C}/pmc_counter_collection.csv')):
            a = agg[clean(r['Kernel_Name'])]; a[0] += 1; a[1] += float(r['Counter_Value'])
    except FileNotFoundError:
        pass
    out[C] = agg
L = [f"# ALIKED, 8 tiles of 1500 x 1000 RGB per extract call (scripts/gpu_aliked_one.py 8 1000 1500), rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes)",
     "# KiB per dispatch summed over the dispatches of ONE extract call sequence (2 calls profiled: warm-up + 1); fetch_x2 = FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md)",
     "%-58s %6s %14s %14s" % ("kernel", "calls", "fetch_x2_MiB", "write_MiB")]
tot_f = tot_w = 0.0
for n in sorted(out['FETCH_SIZE'], key=lambda n: -(2 * out['FETCH_SIZE'][n][1] + out['WRITE_SIZE'].get(n, [0, 0.0])[1])):
    f = out['FETCH_SIZE'][n]; w = out['WRITE_SIZE'].get(n, [1, 0.0])
    L.append("%-58s %6d %14.1f %14.1f" % (n[:58], f[0], 2 * f[1] / 1024, w[1] / 1024)); tot_f += 2 * f[1] / 1024; tot_w += w[1] / 1024
calls = 2
L.append("%-58s %6s %14.1f %14.1f   (per extract call of 8 tiles: %.2f GB; %.3f GB per 1024^2-equivalent image)" %
         ("TOTAL", "", tot_f, tot_w, (tot_f + tot_w) / 1024 / calls, (tot_f + tot_w) / 1024 / calls / 8 / (1500 * 1000 / 1024 ** 2)))
open(f'profiles/{rnd}_aliked_pmc_hbm_summary.txt', 'w').write('\n'.join(L) + '\n')
try:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float); seen = set()
    for r in csv.DictReader(open(f'gpurun_out/pmc_aliked_{tag}_MFMA/pmc_counter_collection.csv')):
        n = clean(r['Kernel_Name']); agg[n][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id']); cnt[n] += 1; dur[n] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    M = ["# rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES, ALIKED 8 x 1500 x 1000",
         "%-58s %6s %10s %9s %9s %12s" % ("kernel", "calls", "avg_us", "clock_GHz", "mfma_busy", "VALU_per_MFMA")]
    for n in sorted(agg, key=lambda n: -dur[n])[:20]:
        a = agg[n]; c = cnt[n]; us = dur[n] / c / 1e3; cyc = a['GRBM_GUI_ACTIVE'] / c / 8
        M.append("%-58s %6d %10.1f %9.2f %9.2f %12s" % (n[:58], c, us, cyc / us / 1e3 if us else 0, a['SQ_VALU_MFMA_BUSY_CYCLES'] / c / (1024 * cyc) if cyc else 0,
                 ("%.1f" % (a['SQ_INSTS_VALU'] / a['SQ_INSTS_MFMA'])) if a['SQ_INSTS_MFMA'] else "-"))
    open(f'profiles/{rnd}_aliked_pmc_mfma_summary.txt', 'w').write('\n'.join(M) + '\n')
except FileNotFoundError:
    pass
print('\n'.join(L[:12])); print("kernel time per extract call of 8 tiles: %.2f ms (rocprofv3 stats over %d profiled calls)" % (total_ms / 4, 4))
