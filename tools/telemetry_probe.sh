#!/bin/bash
# What clock / power sources does this GPU box offer?  (bench.Telemetry picks amdsmi, then hwmon.)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "== hwmon"; for d in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $d; ls $d | tr '\n' ' '; echo; for f in freq1_input power1_average power1_input power1_cap; do [ -e $d/$f ] && echo "$f=$(cat $d/$f)"; done; done
echo "== pp_dpm_sclk"; cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head -8
echo "== cpus"; nproc; python -c "import os; print(os.cpu_count(), len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null
python - <<'PY'
import time, bench
t = bench.Telemetry(0, period_s=0.002)
print("source:", t.source)
if t._read:
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 0.2:
        v = t._read(); n += 1
    print("reads in 0.2 s:", n, "last:", v)
try:
    import amdsmi
    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[0]
    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
    print({k: m[k] for k in ("current_gfxclk", "current_gfxclks", "current_socket_power", "average_socket_power", "average_gfxclk_frequency") if k in m})
    print("bdf", amdsmi.amdsmi_get_gpu_device_bdf(h))
except Exception as e:
    print("amdsmi:", type(e).__name__, e)
PY
