mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sg -o sg -- python3 $GRAFT_REPO_ROOT/tools/sa_group_bench.py 16 > /dev/null 2>&1
python3 - <<'P' > $GRAFT_REPO_ROOT/gpurun_out/r6/c09_sa_group_kernels.txt
import csv, glob, collections
f = glob.glob('/tmp/sg/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0]
    if any(s in k for s in ('sa_group', 'sa_pack', 'ball_query')):
        d[(k, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    v = sorted(v)
    print(f"{k[0][:40]:40s} grid=({k[1]},{k[2]},{k[3]}) n={len(v):3d} median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f}")
P
cat $GRAFT_REPO_ROOT/gpurun_out/r6/c09_sa_group_kernels.txt
