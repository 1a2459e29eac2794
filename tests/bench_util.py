"""bench.py as the tests run it: the ONE compact line from stdout and the detail file it names."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
LINE_LIMIT = 8192      # bytes; what the driver's record is known to take (round 4: 10.9 KB parsed, round 5: 20 KB did not)


class BenchRun:
    def __init__(self, proc, lines, detail):
        self.returncode, self.stdout, self.stderr = proc.returncode, proc.stdout, proc.stderr
        self.lines = lines
        self.line = json.loads(lines[-1]) if lines else None
        self.detail = detail


def run_bench(args, env_extra=None, timeout=900, clean_env=True):
    env = dict(os.environ)
    if clean_env:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
    env.update(env_extra or {})
    with tempfile.TemporaryDirectory(prefix="scl_bench_") as d:
        path = os.path.join(d, "detail.json")
        proc = subprocess.run([sys.executable, BENCH] + list(args) + ["--detail", path], capture_output=True, text=True, env=env,
                              timeout=timeout)
        lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
        detail = None
        if os.path.exists(path):
            with open(path) as fh:
                detail = json.load(fh)
    return BenchRun(proc, lines, detail)


def assert_compact(run):
    """the contract of the line itself: exactly one JSON line on stdout, short enough for the driver's record, parseable"""
    assert len(run.lines) == 1, (run.stdout[-2000:], run.stderr[-2000:])
    assert run.stdout.strip() == run.lines[0]               # nothing else on stdout
    assert len(run.lines[0].encode()) < LINE_LIMIT, len(run.lines[0])
    return run.line
