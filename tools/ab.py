"""A/B of environment-switched variants on one GPU box: runs bench.py once per variant per round, rounds
interleaved, and reports min/median ms per step for every variant (box-to-box and run-to-run noise is ~1 %)."""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    var, values = sys.argv[1], sys.argv[2].split(",")
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    res = {v: [] for v in values}
    for _ in range(rounds):
        for v in values:
            env = dict(os.environ, **{var: v})
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3",
                                  "--no-cpu-baseline", "--no-inference"], env=env, capture_output=True, text=True).stdout
            line = [l for l in out.splitlines() if l.startswith("{")][-1]
            res[v].append(json.loads(line)["ms_per_step"])
    for v in values:
        r = res[v]
        print("%s=%s  min %.3f  median %.3f  all %s" % (var, v, min(r), statistics.median(r), " ".join("%.2f" % x for x in r)), flush=True)


if __name__ == "__main__":
    main()
