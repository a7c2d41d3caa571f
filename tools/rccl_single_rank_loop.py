"""N fresh starts of bench.py over the ONE-rank RCCL group (SHINEON_SINGLE_RANK_GROUP=1: every collective of the exchange path
issued for real, capture in thread-local mode next to ProcessGroupNCCL's watchdog thread), each as its own process; tallies
exit codes.  VERDICT r03 item 7: the thread-local capture fix had been checked on 8 starts only.

    python tools/rccl_single_rank_loop.py [N=30] [out.txt] [-- extra bench args]
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
n = int(args[0]) if args else 30
out = args[1] if len(args) > 1 else None
env = dict(os.environ, SHINEON_SINGLE_RANK_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY="0", SHINEON_DIST_TIMEOUT_S="120")
lines, bad = [], 0
for k in range(n):
    t0 = time.time()
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                            "--no-hbm-table"] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           timeout=180, start_new_session=True)
        rc, err = p.returncode, p.stderr
    except subprocess.TimeoutExpired as e:
        rc, err = -9, (e.stderr or b"").decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
    ok = rc == 0
    bad += 0 if ok else 1
    lines.append(f"start {k + 1:2d}: rc={rc} {time.time() - t0:5.1f} s" + ("" if ok else "  <-- " + err.strip().splitlines()[-1][:200] if err.strip() else ""))
    print(lines[-1], flush=True)
lines.append(f"single-rank RCCL starts ({' '.join(extra) or 'c4 defaults'}): {n - bad} of {n} ok, {bad} failed")
print(lines[-1])
if out:
    with open(out, "a") as f:
        f.write("\n".join(lines) + "\n")
sys.exit(1 if bad else 0)
