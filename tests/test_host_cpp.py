"""hip::net_hip (the C++ drop-in for fpga::net_fpga) driven through net::net_abstract* by a small C++
program, tests/cpp/test_net_hip.cpp.  `cpu` = host-only behaviour (runs here), `gpu` = launch_forward
in MLP and ViT mode against the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_net_hip")


def build_exe():
    pkg = os.path.join(ROOT, "vit-fpga_amd")
    if not (os.path.exists(os.path.join(pkg, "libnetHIP.a")) and os.path.exists(os.path.join(pkg, "libvithip.so"))):
        subprocess.check_call(["make", "-C", pkg, "-j", "4", "all"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    src = os.path.join(ROOT, "tests", "cpp", "test_net_hip.cpp")
    deps = [src, os.path.join(pkg, "libnetHIP.a"), os.path.join(pkg, "host", "netHIP.h")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return
    subprocess.check_call(["g++", "-std=gnu++14", "-O1", "-Wall", f"-I{pkg}/host", f"-I{ROOT}/include", src, "-o", EXE,
                           f"{pkg}/libnetHIP.a", f"-L{pkg}", "-lvithip", f"-L{ROOT}/oracle", "-loracle",
                           "-Wl,-rpath,$ORIGIN/../../vit-fpga_amd", "-Wl,-rpath,$ORIGIN/../../oracle"])


def run(mode):
    build_exe()
    p = subprocess.run([EXE, mode], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert f"{mode}: 0 failure(s)" in p.stdout


def test_net_hip_host_behaviour():
    run("cpu")


def test_reference_style_caller_compiles_against_our_headers(tmp_path):
    # a translation unit written the way a user of the reference writes one: includes by bare name,
    # uses every virtual through the abstract base, constructs with the reference's ctor signature
    src = tmp_path / "caller.cpp"
    src.write_text('''
#include <netHIP.h>
#include <memory>
int use(net::net_abstract &n, const net::net_sets &s, const net::image_set &im) {
    net::net_data d = n.get_net_data();
    std::vector<DATA_TYPE> y = n.launch_forward(std::vector<DATA_TYPE>(d.n_ins, net::MAX_RANGE));
    n.init_gradient(s);
    std::vector<DATA_TYPE> e = n.launch_gradient(3, DATA_TYPE(0.1), DATA_TYPE(0.5));
    n.print_inner_vals();
    signed long a = n.get_gradient_performance() + n.get_forward_performance();
    n.filter_image(im);
    net::image_set o = n.get_filtered_image();
    return (int)(y.size() + e.size() + a + o.original_h);
}
net::net_abstract *make(const net::net_data &d) { return new hip::net_hip(d, false, true); }
''')
    pkg = os.path.join(ROOT, "vit-fpga_amd")
    subprocess.check_call(["g++", "-std=gnu++14", "-fsyntax-only", "-Wall", "-Werror", f"-I{pkg}/host",
                           f"-I{ROOT}/include", str(src)])


@pytest.mark.gpu
def test_net_hip_forward_on_gpu():
    run("gpu")
