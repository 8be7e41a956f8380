set -e
cd /root/repo
export TMPDIR=/tmp
L=gpurun_out/exp_iter_split.log
: > $L
timeout -k 10 300 python tools/ab.py compact_debug=0,16,0,16,0,16 >> $L 2>&1
timeout -k 10 300 python tools/ab.py --mesh bwr_like.msh --nazim 64 --delta 2e-3 compact_debug=0,16,0,16 >> $L 2>&1
timeout -k 10 500 python tools/ab.py --nohash --mesh bwr_like.msh --nazim 128 --delta 5e-4 compact_debug=0,16,0,16 >> $L 2>&1
grep -v amdgpu.ids $L | sed 's/records sha \([0-9a-f]*\).*/sha \1/; s/ volumes 0.0000//; s/in-tree | //' | cut -c1-330
