"""Build the committed profile summaries (profiles/) from the rocprofv3 outputs merged into gpurun_out/."""
import csv, collections, json, re, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
images_per_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 32  # bench.py default: 16 pairs per step = 32 images per conv launch
def clean(n, grid=None):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); n = n.split('(')[0].replace(', ', ',')
    # the conv / GEMM kernels serve several layers: one line per launch shape (grid = workgroups x 256 threads)
    if grid is not None and (n.startswith('conv3x3_x6') or n.startswith('gemm_x6')): n += ' g' + str(int(grid) // 256)
    return n
shutil.copy(f'gpurun_out/prof_{tag}/bench_kernel_stats.csv', f'profiles/{tag}_bench_kernel_stats.csv')
shutil.copy(f'gpurun_out/bench_{tag}.json', f'profiles/{tag}_bench_n1.json')
out = {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.defaultdict(lambda: [0, 0.0, '', '', ''])
    for r in csv.DictReader(open(f'gpurun_out/pmc_{tag}_{C}/pmc_counter_collection.csv')):
        a = agg[clean(r['Kernel_Name'], r.get('Grid_Size'))]; a[0] += 1; a[1] += float(r['Counter_Value']); a[2] = r['VGPR_Count']; a[3] = r['LDS_Block_Size']; a[4] = r['SGPR_Count']
    out[C] = agg
names = sorted(out['FETCH_SIZE'], key=lambda n: -out['FETCH_SIZE'][n][1])
L = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) on `python bench.py --steps 2 --warmup 1 --no-cpu-baseline`",
     "# counter unit: KiB per dispatch (WRITE_SIZE checks out exactly against known byte counts: e.g. the pooled conv1b map of 32 images is 32 x 67,108,864 B = 2,097,152 KiB).",
     "# gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads -> 'fetch_x2' column.",
     "# conv / GEMM lines are per launch shape: g<workgroups>; at 100 images of 1024^2: conv1b g204800, conv2a / conv2b g51200, conv3a g25600 (<64,0>), conv3b g25600 (<128,1>),",
     "# conv4a / conv4b g6400, convPa / convDa g12800; GEMM g12800 = SuperPoint's 1x1 heads, g1600 / g3200 / g4800 = LightGlue's 256- / 512- / 768-column linears.",
     "%-66s %6s %14s %14s %14s %6s %7s %5s" % ("kernel", "calls", "fetch_KiB_avg", "fetch_x2_KiB", "write_KiB_avg", "vgpr", "lds_B", "sgpr")]
for n in names[:34]:
    f = out['FETCH_SIZE'][n]; w = out['WRITE_SIZE'].get(n, [1, 0.0])
    L.append("%-66s %6d %14.1f %14.1f %14.1f %6s %7s %5s" % (n[:66], f[0], f[1] / f[0], 2 * f[1] / f[0], w[1] / max(1, w[0]), f[2], f[3], f[4]))
open(f'profiles/{tag}_pmc_hbm_summary.txt', 'w').write('\n'.join(L) + '\n')
k = [n for n in names if n.startswith('conv3x3_x6_kernel<64,1,1,true,2')][0]
f = out['FETCH_SIZE'][k]; w = out['WRITE_SIZE'][k]
hb = (2 * f[1] / f[0] + w[1] / w[0]) * 1024
json.dump({"kernel": k, "hbm_bytes_per_launch": hb, "fetch_bytes_corrected_x2": 2 * f[1] / f[0] * 1024, "write_bytes": w[1] / w[0] * 1024,
           "source": f"profiles/{tag}_pmc_hbm_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM)",
           "batch_images_per_launch": images_per_launch}, open('profiles/conv1b_hbm_bytes.json', 'w'), indent=1)
# MFMA / clock summary
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float); seen = set()
for r in csv.DictReader(open(f'gpurun_out/pmc_{tag}_MFMA/pmc_counter_collection.csv')):
    n = clean(r['Kernel_Name'], r.get('Grid_Size')); agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); cnt[n] += 1; dur[n] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
L = ["# rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES on `python bench.py --steps 2 --warmup 1`",
     "# clock_GHz = GRBM_GUI_ACTIVE / 8 XCDs / duration; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles); waves/SIMD = 4 x SQ_WAVE_CYCLES / (1024 x cycles)",
     "%-66s %6s %10s %9s %9s %10s %12s" % ("kernel", "calls", "avg_us", "clock_GHz", "mfma_busy", "waves/SIMD", "VALU_per_MFMA")]
for n in sorted(agg, key=lambda n: -dur[n])[:24]:
    a = agg[n]; c = cnt[n]; us = dur[n] / c / 1e3; cyc = a['GRBM_GUI_ACTIVE'] / c / 8
    L.append("%-66s %6d %10.1f %9.2f %9.2f %10.2f %12s" % (n[:66], c, us, cyc / us / 1e3, a['SQ_VALU_MFMA_BUSY_CYCLES'] / c / (1024 * cyc) if cyc else 0,
             4 * a['SQ_WAVE_CYCLES'] / c / (1024 * cyc) if cyc else 0, ("%.1f" % (a['SQ_INSTS_VALU'] / a['SQ_INSTS_MFMA'])) if a['SQ_INSTS_MFMA'] else "-"))
open(f'profiles/{tag}_pmc_mfma_summary.txt', 'w').write('\n'.join(L) + '\n')
print('\n'.join(L)); print(hb / 1e9, 'GB/launch for', k)
