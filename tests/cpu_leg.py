"""The CPU leg of the measurement: cpu::net_cpu (tests/cpp/net_cpu.{h,cpp}: the oracle behind net::net_abstract) timed
through launch_forward by tests/cpp/cpu_leg.cpp.  TEST INFRASTRUCTURE: used by bench.py's cpu_baseline and by tests."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
EXE = os.path.join(CPP, "cpu_leg")


def build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    srcs = [os.path.join(CPP, f) for f in ("cpu_leg.cpp", "net_cpu.cpp", "net_cpu.h")] + [os.path.join(ROOT, "oracle", "liboracle.so")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(s) for s in srcs):
        return EXE
    host = os.path.join(ROOT, "vit-fpga_amd", "host")   # netAbstract.h / defines.h (the re-declared plugin interface)
    subprocess.check_call(["g++", "-std=gnu++14", "-O2", "-Wall", f"-I{host}", os.path.join(CPP, "cpu_leg.cpp"),
                           os.path.join(CPP, "net_cpu.cpp"), "-o", EXE, f"-L{ROOT}/oracle", "-loracle", "-fopenmp",
                           "-Wl,-rpath,$ORIGIN/../../oracle"])
    return EXE


def run(cfg, batch, threads, max_seconds):
    exe = build()
    args = [exe] + [str(cfg[k]) for k in ("image_size", "patch_size", "channels", "dim", "heads", "mlp_dim", "layers", "classes")]
    args += [str(batch), str(threads), str(max_seconds)]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    p = subprocess.run(args, capture_output=True, text=True, timeout=max(120, 6 * max_seconds), env=env)
    if p.returncode != 0:
        raise RuntimeError(f"cpu_leg failed: {p.stderr}")
    return json.loads(p.stdout.strip().splitlines()[-1])


def baseline(cfg, cores, target_seconds):
    """ViT-Tiny/16 batch 1 and the benchmarked model (a bounded sample: batch 32, or fewer images
    when a forward would not fit the budget), at 1 thread and at all granted threads; median over >= 5 runs."""
    import vh_synth as S
    tiny = S.CONFIGS["vit_tiny"]
    share = target_seconds / 4.0
    legs = {}
    legs["vit_tiny_b1_threads1"] = run(tiny, 1, 1, share * 0.5)
    legs[f"vit_tiny_b1_threads{cores}"] = run(tiny, 1, cores, share * 0.5)
    big_b = 32
    main = run(cfg, big_b, cores, share * 1.5)
    legs[f"model_b{big_b}_threads{cores}"] = main
    legs["model_b2_threads1"] = run(cfg, 2, 1, share * 1.5)
    return {"value": round(main["images_per_s"], 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{big_b} images of the same workload per launch_forward, median of {main['runs']} runs, "
                      f"{cores} OpenMP threads; cpu::net_cpu (the fp32 oracle behind net::net_abstract, tests/cpp/net_cpu) "
                      "timed with the reference's chrono window inside launch_forward (netFPGA.cpp:262-284); "
                      "the reference itself ships no CPU path",
            "legs": {k: {"images_per_s": round(v["images_per_s"], 3), "median_us": v["median_us"], "runs": v["runs"],
                         "batch": v["batch"], "threads": v["threads"]} for k, v in legs.items()}}
