"""What the driver does with bench.py's output, as a test helper: the LAST line of the last 8 000 bytes of stdout must be one JSON
object (BENCH_r05.json came back `parsed: null` on a 23.6 KB line).  bench.py writes that compact line to stdout
and nothing else; the detail object goes to bench_detail.json and, tagged, to stderr."""
import json

LINE_LIMIT = 8000
DETAIL_TAG = "bench_detail "
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline", "ranks")


def parse(stdout, stderr=""):
    """-> (compact line as the driver would recover it, detail object or None); ([], None) when bench.py printed no line"""
    rows = [l for l in stdout.splitlines() if l.strip()]
    if not rows:
        return None, None
    # (the gloo transport of the CPU test harness announces its ranks on stdout; bench.py itself writes the one line)
    assert len([r for r in rows if r.startswith("{")]) == 1, "stdout carries ONE JSON object, the compact line: %r" % [r[:80] for r in rows]
    last = rows[-1]
    assert last.startswith("{") and len(last) < LINE_LIMIT, len(last)
    tail = stdout[-LINE_LIMIT:]
    line = json.loads([l for l in tail.splitlines() if l.strip()][-1])          # what a reader of the 8 KB tail recovers
    for k in CONTRACT:
        assert k in line, k
    assert "workload" in line["config"] and "model" not in line["config"]
    tagged = [l for l in stderr.splitlines() if l.startswith(DETAIL_TAG)]
    detail = json.loads(tagged[-1][len(DETAIL_TAG):]) if tagged else None
    if detail is not None:       # the compact line is a projection of the detail object, not a second measurement
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling"):
            assert line[k] == detail[k], k
        if detail.get("roofline"):
            for k in ("kernel_ms", "frac", "achieved", "peak", "traffic", "executed_macs_per_unit"):
                assert line["roofline"][k] == detail["roofline"][k], k
        assert line["ranks"]["world_size"] == detail["ranks"]["world_size"]
    return line, detail
