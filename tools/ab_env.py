"""A/B of run-time knobs of the library (GPU box): the aggregator's stage times on the bench workload, one process per
environment setting.    python tools/ab_env.py PN_SEQ_STEP=0 PN_SEQ_STEP=1 [steps=10]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tune_run import CHILD, ROOT  # noqa: E402


def main():
    steps, specs = 10, []
    for a in sys.argv[1:]:
        if a.startswith("steps="):
            steps = int(a[6:])
        else:
            specs.append(a)
    for spec in specs or [""]:
        env = dict(os.environ)
        env.update(kv.split("=", 1) for kv in spec.split(",") if kv)
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, steps)], env=env, capture_output=True, text=True)
        res = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        print("%-40s %s" % (spec or "(default)", res[0][7:] if res else "FAILED " + r.stderr[-800:]), flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
