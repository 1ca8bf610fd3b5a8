"""The `-m gpu` parity tests, on the CPU tier, against the HOST-EMULATED library.

tests/hostemu/build.py compiles the library's own sources — capi.cpp, engine.cpp, elementwise / norm / gemm / attention .hip — as plain C++
against a stand-in HIP runtime: "device" memory is host memory, a launch runs its blocks one after the other with every GPU thread a fiber,
the gfx950 matrix / permute / LDS-DMA builtins are emulated over a wave's 64 lanes.  The result exports the whole C ABI of include/sdmi.h.
A subprocess then runs the GPU parity tests themselves — the files tests/test_gpu_*.py, unchanged — with that library loaded through
SDMI_LIB and torch's `cuda` device mapped to the CPU for the test process (tests/hostemu/shim.py): the product's Python host code, its
ctypes binding, the engine's C++ and every kernel's source execute as they do on the MI355X, compared with the same oracle to the same
tolerances.  What this tier cannot see is what needs the hardware: timing, the asynchrony of LDS-DMA and of streams, the real matrix
cores' rounding of products (emulated as fp32 multiply-adds in k order, which the parity on the GPU shows to be within the same tolerances).

The selection (tests/hostemu/cpu_tier_selection.txt) is every GPU test that takes at most 15 s there; all other GPU tests that finish
inside a minute under emulation pass too.  Nothing here is linked into, or imported by, the product.
"""
import ctypes as C
import importlib
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECTION = os.path.join(ROOT, "tests", "hostemu", "cpu_tier_selection.txt")


def test_emulated_library_exports_the_whole_c_abi(hostemu_lib, monkeypatch):
    lib = C.CDLL(hostemu_lib)
    declared = importlib.import_module("stable-diffusion-webui_amd._lib").declared_symbols()
    assert len(declared) > 60
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    lib.sdmi_device_ok.restype = C.c_int
    monkeypatch.delenv("SDMI_HOSTEMU", raising=False)
    assert lib.sdmi_device_ok() == 0                           # pointed at by SDMI_LIB alone it is NOT a device: the product's require_device() refuses
    monkeypatch.setenv("SDMI_HOSTEMU", "1")
    assert lib.sdmi_device_ok() == 1                           # only a test process that asked for the emulation sees a gfx950 "device"


def test_gpu_parity_tests_pass_on_the_emulated_library(hostemu_lib):
    selected = [ln.strip() for ln in open(SELECTION) if ln.strip() and not ln.startswith("#")]
    assert len(selected) > 100
    env = dict(os.environ, SDMI_HOSTEMU="1", SDMI_LIB=hostemu_lib, SDMI_HOSTEMU_SELECT=SELECTION)
    env.pop("PYTEST_CURRENT_TEST", None)
    workers = str(max(1, min(8, os.cpu_count() or 1)))
    cmd = [sys.executable, "-m", "pytest", "tests/test_gpu_ops.py", "tests/test_gpu_models.py", "tests/test_gpu_boundaries.py", "-m", "gpu", "-q",
           "-p", "no:cacheprovider", "-n", workers, "--timeout=300", "--timeout-method=thread"]
    run = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    out = run.stdout.decode(errors="replace")
    tail = out[-6000:]
    assert run.returncode == 0, tail
    m = re.search(r"(\d+) passed", out)
    assert m, tail
    # (the selection file names tests by node id; a renamed test drops out of it rather than failing the run: at least 90 % must still be there)
    assert int(m.group(1)) >= 0.9 * len(selected), tail
    assert "failed" not in out.splitlines()[-1], tail


def _run_emulated(args, hostemu_lib, extra_env=None, timeout=900):
    env = dict(os.environ, SDMI_HOSTEMU="1", SDMI_LIB=hostemu_lib, **(extra_env or {}))
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "hostemu", "run.py")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE)


BENCH_SMALLEST = ["--model", "tiny", "--size", "64", "--sampler-steps", "2", "--steps", "1", "--warmup", "1", "--no-pmc-traffic"]


@pytest.fixture(scope="module")
def emulated_runs(hostemu_lib):
    """smoke(), bench.py on one rank and bench.py on two gloo ranks, started together (four processes) and collected once."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = {"smoke": _run_emulated(["-c", "import __graft_entry__ as g; g.smoke(); print('smoke ok')"], hostemu_lib),
             "bench1": _run_emulated(["bench.py", "--batch", "1"] + BENCH_SMALLEST, hostemu_lib)}
    args = ["bench.py", "--gpus", "2", "--batch", "2", "--no-roofline", "--no-cpu-baseline", "--verify-shards"] + BENCH_SMALLEST
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "WORLD_SIZE": "2", "LOCAL_WORLD_SIZE": "2", "SDMI_DIST_BACKEND": "gloo", "LOCAL_RANK": "0"}
    for r in range(2):
        procs[f"rank{r}"] = _run_emulated(args, hostemu_lib, dict(env, RANK=str(r)))
    res = {}
    for name, pr in procs.items():
        out, err = pr.communicate(timeout=900)
        res[name] = (pr.returncode, out.decode(errors="replace"), err.decode(errors="replace")[-3000:])
    return res


def test_smoke_entry_on_the_emulated_library(emulated_runs):
    """__graft_entry__.smoke() — one small txt2img job checked against the oracle — as the driver calls it, minus the GPU."""
    rc, out, err = emulated_runs["smoke"]
    assert rc == 0 and "smoke ok" in out, err


def test_bench_py_end_to_end_on_the_emulated_library(emulated_runs):
    """bench.py itself, one rank, tiny model: the timed loop, the per-launch profile behind the roofline block (launch names from the
    emulation's profiler stand-in; its times are host time and mean nothing), the CPU-baseline leg, the one JSON line of the contract."""
    import json
    rc, out, err = emulated_runs["bench1"]
    assert rc == 0, err
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 1 and line["value"] > 0 and line["unit"] == "images/s" and line["vs_baseline"] is None
    assert line["roofline"]["bound"] == "mfma" and line["roofline"]["launches_per_job"] > 50 and line["roofline"]["dominant_variant"]["name"].startswith("gemm_mfma_")
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0 and "workload" in line["config"]


def test_bench_py_two_ranks_over_gloo_on_the_emulated_library(emulated_runs):
    """The N > 1 path of bench.py with the real engine on both ranks (tests/test_gpu_boundaries.py's two-rank case, minus the GPU): weights
    generated on rank 0 and broadcast, the job sharded by process_images_sharded, uint8 images gathered on rank 0, barrier + max over ranks,
    and --verify-shards: every rank's slice replayed on rank 0 equals the gathered images bit for bit."""
    import json
    (rc0, out0, err0), (rc1, out1, err1) = emulated_runs["rank0"], emulated_runs["rank1"]
    assert rc0 == 0 and rc1 == 0, (err0, err1)
    line = json.loads([ln for ln in out0.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak"
    assert line["config"]["shard_check"] == "ok", line["config"]["shard_check"]
    assert line["config"]["weights_broadcast_ms"] > 0 and line["value"] > 0
    assert not [ln for ln in out1.splitlines() if ln.startswith("{")]           # only rank 0 prints the line
