"""Condense rocprofv3 --pmc output (one sub-directory per pass) into a per-kernel table: the average of every
counter over the dispatches of each kernel whose name contains the filter, plus a few derived figures.
usage: python tools/pmc_summary.py <dir> [name filter]"""
import csv, glob, os, re, sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
acc = defaultdict(lambda: defaultdict(list))       # kernel -> counter -> values
dur = defaultdict(list)
meta = {}
csv.field_size_limit(1 << 30)
for f in sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if flt not in name:
            continue
        short = re.sub(r'\(.*', '', name).replace('void ', '')
        acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[short].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        meta[short] = (r['Grid_Size'], r['Workgroup_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'])
for k in acc:
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    us = sum(dur[k]) / len(dur[k])
    print(f'== {k}')
    print(f'   grid {meta[k][0]} wg {meta[k][1]} vgpr {meta[k][2]} agpr {meta[k][3]} lds {meta[k][4]}; avg duration under the profiler {us:.1f} us')
    for n in sorted(c):
        print(f'   {n:36s} {c[n]:18.0f}')
    g = c.get
    if g('SQ_WAVE_CYCLES'):
        w = g('SQ_WAVE_CYCLES')
        print(f"   -- wave-cycle shares: WAIT_ANY {g('SQ_WAIT_ANY', 0) / w:.3f}  WAIT_INST_ANY {g('SQ_WAIT_INST_ANY', 0) / w:.3f}  "
              f"ACTIVE_INST_ANY {g('SQ_ACTIVE_INST_ANY', 0) / w:.3f}  WAIT_INST_LDS {g('SQ_WAIT_INST_LDS', 0) / w:.3f}")
    if g('SQ_VALU_MFMA_BUSY_CYCLES') and g('SQ_BUSY_CYCLES'):
        # SQ_BUSY_CYCLES counts per SE (32 of them); MFMA busy counts cycles per SIMD summed: 1024 SIMDs
        print(f"   -- MFMA busy / (4 SIMD x CU busy cycles): {g('SQ_VALU_MFMA_BUSY_CYCLES') / (4.0 * g('SQ_BUSY_CU_CYCLES', 0) or 1):.3f} (needs SQ_BUSY_CU_CYCLES)"
              if g('SQ_BUSY_CU_CYCLES') else f"   -- MFMA busy cycles per SIMD (of 1024): {g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024:.0f}")
    if g('SQ_LDS_IDX_ACTIVE'):
        print(f"   -- LDS bank-conflict share of LDS-array cycles: {g('SQ_LDS_BANK_CONFLICT', 0) / g('SQ_LDS_IDX_ACTIVE'):.4f}")
    if g('TCC_HIT_sum') is not None and g('TCC_MISS_sum') is not None and g('TCC_HIT_sum', 0) + g('TCC_MISS_sum', 0) > 0:
        print(f"   -- L2 hit rate: {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.4f}")
    if g('TCP_TCC_READ_REQ_sum'):
        print(f"   -- mean L1->L2 read latency: {g('TCP_TCC_READ_REQ_LATENCY_sum', 0) / g('TCP_TCC_READ_REQ_sum'):.0f} cycles")
    if g('GRBM_GUI_ACTIVE'):
        print(f"   -- effective clock: {g('GRBM_GUI_ACTIVE') / 8 / us / 1e3:.2f} GHz (GRBM_GUI_ACTIVE / 8 / duration)")
    if g('FETCH_SIZE') is not None:
        print(f"   -- HBM-side read bytes: 2 x FETCH_SIZE KB = {2 * g('FETCH_SIZE') * 1024 / 1e6:.1f} MB" + (f"; written {g('WRITE_SIZE') * 1024 / 1e6:.1f} MB" if g('WRITE_SIZE') is not None else ''))
