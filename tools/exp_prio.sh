# round 5: the record kernel's header priority on / off (option "compact_debug" 8 = off), same box (development)
set -e
cd /root/repo
export TMPDIR=/tmp
L=gpurun_out/exp_prio_ab.log
: > $L
timeout -k 10 300 python tools/ab.py compact_debug=8,0,8,0 >> $L 2>&1
timeout -k 10 300 python tools/ab.py --mesh bwr_like.msh --nazim 64 --delta 2e-3 compact_debug=8,0,8,0 >> $L 2>&1
timeout -k 10 500 python tools/ab.py --nohash --mesh bwr_like.msh --nazim 128 --delta 5e-4 compact_debug=8,0,8,0 >> $L 2>&1
grep -v amdgpu.ids $L
