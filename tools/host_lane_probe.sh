# tools/host_lane_probe.sh -- host CLI lanes on one GPU box: where does the time go when lanes run side by side? (AIM_HOST_DRY = host-side only)
cd "$(dirname "$0")/.."
python tools/cli_scale.py 64 --only packed --keep > /dev/null 2>&1
N=67108864
H=aim_amd/host/host
C="$N --algo wfa --max-score 5 --read-size 112 --reduce"
cat /proc/loadavg; echo "numa_balancing $(cat /proc/sys/kernel/numa_balancing 2>/dev/null) thp $(cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null)"; free -g | head -2
vm() { grep -E "^(numa_hint_faults|numa_pages_migrated|pgmigrate_success|pgfault|thp_fault_alloc|numa_pte_updates) " /proc/vmstat | tr '\n' ' '; echo; }
one() { echo "== $1"; shift; vm; AIM_HOST_RUSAGE=1 "$@" 2>&1 | tail -2 | sed -e 's/^AIM-HIP: //' -e 's/over 1 device(s) x 2 slot(s); //' -e 's/input packed.*lane(s) x/ lanes x/'; vm; }
for k in 1 4; do for f in $(seq 0 $((k-1))); do ln -sf /dev/null /tmp/n.out.$(printf %03d $f); done; done
ln -sf /dev/null /tmp/n.out
one "text 1 lane dry null" env AIM_HOST_DRY=1 $H /tmp/aim_scale_64x.seq /tmp/n.out $C
one "text 1 lane dry null 96+32 threads" env AIM_HOST_DRY=1 $H /tmp/aim_scale_64x.seq /tmp/n.out $C --pack-threads 96 --format-threads 32
one "text 4 lanes dry null pinned" env AIM_HOST_DRY=1 $H /tmp/aim_scale_64x.seq /tmp/n.out $C --out-shards 4
one "text 4 lanes dry null no-pin" env AIM_HOST_DRY=1 $H /tmp/aim_scale_64x.seq /tmp/n.out $C --out-shards 4 --no-pin
one "text 4 lanes dry null no-pin 16+8" env AIM_HOST_DRY=1 $H /tmp/aim_scale_64x.seq /tmp/n.out $C --out-shards 4 --no-pin --pack-threads 16 --format-threads 8
one "text 4 lanes dry null pinned 16+8" env AIM_HOST_DRY=1 $H /tmp/aim_scale_64x.seq /tmp/n.out $C --out-shards 4 --pack-threads 16 --format-threads 8
one "text 4 lanes dry null pinned no-populate" env AIM_HOST_DRY=1 AIM_HOST_POPULATE=0 $H /tmp/aim_scale_64x.seq /tmp/n.out $C --out-shards 4
one "packed 4 lanes dry null pinned" env AIM_HOST_DRY=1 $H /tmp/aim_scale_64x.aimpk /tmp/n.out $C --out-shards 4 --packed-input
one "packed 1 lane dry null" env AIM_HOST_DRY=1 $H /tmp/aim_scale_64x.aimpk /tmp/n.out $C --packed-input
cat /proc/loadavg
