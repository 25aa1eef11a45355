"""Serial timeline of the sparse-voxel branch inside one eager single-stream step (rocprofv3 kernel trace of
`bench.py --vox --graph 0 --streams 1`): per category totals and the gather-GEMM launches in order.
usage: python3 tools/vox_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
starts = [i for i, n in enumerate(names) if 'keys_kernel' in n or 'keys_hist_kernel' in n]
i0, i1 = starts[-2], starts[-1]
cats = {}
order = []
for r in rows[i0:i1]:
    n = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'rocprim' in n: c = 'rocprim sort'
    elif 'zplane_perm' in n or 'tile_taps' in n: c = 'row order + tap sets (zplane_perm, tile_taps)'
    elif 'agp_coords' in n: c = 'coordinate manager (keys / scatter / seg_sort / place)'
    elif 'igemm_kernel' in n or 'spwin' in n: c = 'gather-GEMM'; order.append(d)
    elif 'conv0' in n: c = 'conv0 (125 taps, Cin 1)'
    elif 'kernel_map' in n: c = 'kernel maps'
    elif 'seg_' in n or 'eca' in n: c = 'segment pool / ECA / affine'
    elif 'agp_sparse' in n: c = 'other sparse'
    else: c = 'image path + vector programs + torch'
    t = cats.setdefault(c, [0, 0.0]); t[0] += 1; t[1] += d
span = (int(rows[i1]['Start_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3
for c, (k, d) in sorted(cats.items(), key=lambda kv: -kv[1][1]):
    print(f"{c:50s} {k:4d} launches {d:9.1f} us")
print(f"step span {span:.1f} us; gather-GEMM launches (us): " + ' '.join(f"{d:.0f}" for d in order))
