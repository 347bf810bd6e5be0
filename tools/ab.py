"""A/B runs of bench.py with Python-level switches of rstnet_amd.ops flipped (no environment variable reaches the library):

    python tools/ab.py SKINNY_X32_MAX_K=0 -- --workload gpt --steps 40 --warmup 5 --no-cpu-baseline

Every NAME=VALUE before `--` is set on rstnet_amd.ops (int / float / bool literals), the rest is bench.py's command line.
`LIB=<path>` loads another build of the library instead (the tools build `make -C rstnet_amd/csrc ablation`, whose measurement
knobs read RST_* environment variables; the shipped library reads none)."""
import ast
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    args = sys.argv[1:]
    cut = args.index("--") if "--" in args else len(args)
    from rstnet_amd import _lib, ops
    for item in args[:cut]:
        name, value = item.split("=", 1)
        if name == "LIB":
            _lib.LIB_PATH = os.path.abspath(value)
            continue
        assert hasattr(ops, name), f"rstnet_amd.ops has no switch {name}"
        setattr(ops, name, ast.literal_eval(value))
    sys.argv = [os.path.join(ROOT, "bench.py")] + args[cut + 1:]
    runpy.run_path(sys.argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
