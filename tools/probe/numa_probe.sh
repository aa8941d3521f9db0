# Which NUMA node is the GPU on, and what does the drop-in loop make when its thread is kept on either? (GPU box only.)
ls /sys/devices/system/node/ | grep node
for n in /sys/devices/system/node/node*; do echo "$n $(cat $n/cpulist)"; done
for d in /sys/class/drm/card*/device; do echo "$d numa $(cat $d/numa_node 2>/dev/null)"; done
rocm-smi --showtoponuma 2>/dev/null | head -20
for n in /sys/devices/system/node/node*; do c=$(cat $n/cpulist); echo "cpus $c: $(taskset -c $c python tools/bench_next.py --pairs 24000000 --consume 0 2>&1 | tail -1 | cut -c95-130)"; done
