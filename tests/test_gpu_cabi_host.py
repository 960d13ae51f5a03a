"""The C ABI from a plain C host (no Python, no PyTorch on the calling side): build examples/host_c with gcc and run it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_plain_c_host_projects_through_the_abi(tmp_path):
    gcc = shutil.which('gcc') or 'gcc'
    libdir = os.path.join(ROOT, 'adaptivepnp_sci_amd')
    exe = str(tmp_path / 'gap_projection_host')
    subprocess.run([gcc, '-std=c11', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I', os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'examples', 'host_c', 'gap_projection_host.c'), '-L', libdir, '-lscipnp',
                    '-L/opt/rocm/lib', '-lamdhip64', '-lm', f'-Wl,-rpath,{libdir}', '-Wl,-rpath,/opt/rocm/lib', '-o', exe],
                   check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'projection rel-L2' in r.stdout and 'misaligned call -> -2' in r.stdout
