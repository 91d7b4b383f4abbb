"""Measured parity errors of the bf16-MFMA mode, recorded while the GPU tests run.

Every bf16-mode test calls `record(case, metric=value, ...)` with what it measured before asserting its tolerance; the values
are merged into gpurun_out/r06_parity.json on the GPU box (committed as profiles/r06_parity.json), so a tolerance in a test can be
read next to the error it bounds.  The rule -- a bf16 tolerance is at most 2x the recorded worst case -- is checked mechanically:
`bounds={metric: tolerance}` stores the tolerance the test asserts as "bound.<metric>" beside the measurement, and
tests/test_host_cpu.py::test_bf16_bounds_at_most_twice_the_measured_error reads the committed file."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "gpurun_out", "r06_parity.json")


def record(case: str, bounds=None, **metrics):
    vals = {k: (float(v) if isinstance(v, (int, float)) else v) for k, v in metrics.items()}
    for k, b in (bounds or {}).items():
        assert k in vals, k
        vals["bound." + k] = float(b)
    print(f"[parity] {case}: " + ", ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in vals.items()))
    if os.environ.get("MI_CONV_AUTO", "1") != "1":       # the halo-only subprocess run measures other kernels: printed, not recorded
        return vals
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        data = {}
        if os.path.exists(PATH):
            with open(PATH) as f:
                data = json.load(f)
        data[case] = vals
        with open(PATH, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass
    return vals
