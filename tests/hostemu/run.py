"""python tests/hostemu/run.py <script.py> [args ...]  |  -c "<statements>" — runs a repository script (bench.py, a tools/ script) on a machine WITHOUT a GPU against
the host-emulated library: builds it if needed (tests/hostemu/build.py), points SDMI_LIB at it, installs tests/hostemu/shim.py, then executes
the script as __main__.  TEST INFRASTRUCTURE; the numbers such a run prints are host time of an emulation and mean nothing as performance."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from hostemu import build, shim  # noqa: E402

if __name__ == "__main__":
    if os.environ.get("SDMI_HOSTEMU") != "1" or not os.environ.get("SDMI_LIB"):
        lib = build.build()
        if not lib:
            sys.exit("host emulation needs clang++ on x86-64")
        os.environ["SDMI_LIB"], os.environ["SDMI_HOSTEMU"] = lib, "1"
    shim.install()
    if sys.argv[1] == "-c":                                    # python tests/hostemu/run.py -c "<statements>"
        code, sys.argv = sys.argv[2], ["-c"] + sys.argv[3:]
        exec(compile(code, "<-c>", "exec"), {"__name__": "__main__"})
    else:
        script = sys.argv[1]
        sys.argv = sys.argv[1:]
        sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
        runpy.run_path(script, run_name="__main__")
