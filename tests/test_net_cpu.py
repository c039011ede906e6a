"""cpu::net_cpu (tests/cpp/net_cpu.{h,cpp}): the CPU implementer of net::net_abstract that BASELINE config 1 names — the
oracle behind the reference's plugin interface (test infrastructure; the reference ships no CPU path).  Host only."""
import os
import subprocess

import numpy as np

import cpu_leg
import oracle_lib as O
import vh_synth as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def test_net_cpu_through_the_abstract_interface():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    exe = os.path.join(CPP, "test_net_cpu")
    host = os.path.join(ROOT, "vit-fpga_amd", "host")
    subprocess.check_call(["g++", "-std=gnu++14", "-O1", "-Wall", "-Werror", f"-I{host}", os.path.join(CPP, "test_net_cpu.cpp"),
                           os.path.join(CPP, "net_cpu.cpp"), "-o", exe, f"-L{ROOT}/oracle", "-loracle", "-fopenmp",
                           "-Wl,-rpath,$ORIGIN/../../oracle"])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(p.stdout, p.stderr)
    assert p.returncode == 0 and "net_cpu: 0 failure(s)" in p.stdout


def test_cpu_leg_times_config_1_through_launch_forward():
    """BASELINE config 1: ViT-Tiny/16, one 224x224x3 image, fp32, CPU, through net_abstract::launch_forward; the logits
    are the oracle's (checked by their sum against the Python binding of the same oracle on the same seeded data)."""
    cfg = S.CONFIGS["vit_tiny"]
    r = cpu_leg.run(cfg, 1, 2, 1.0)
    assert r["batch"] == 1 and r["threads"] == 2 and r["runs"] >= 5 and r["images_per_s"] > 0
    ref = O.vit_forward(cfg, S.make_blob(cfg, 0), S.make_images(cfg, 1, 1), threads=2)
    assert abs(r["logit_sum"] - float(ref.astype(np.float64).sum())) <= 1e-4 * max(1.0, np.abs(ref).sum())
