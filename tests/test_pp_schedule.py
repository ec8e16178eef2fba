"""CPU: the LDS-DMA / barrier schedules of the persistent ping-pong kernels (gemm_pp.hip, conv_pp.hip) replayed by
tools/check_pp_schedule.py -- every block waited for (counted vmcnt) one barrier before any wave reads it, and issued into an LDS
region only >= 2 phases after the last read of the block it overwrites, for the two wave groups one barrier apart."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pp_schedules_have_no_raw_or_war_hazard():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_pp_schedule.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    for name in ("pp256", "pp192", "conv_pp128", "pp192_resid"):
        assert f"{name}: OK" in out.stdout, out.stdout
