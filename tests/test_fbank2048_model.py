"""CPU check of the design of fbank2048_kernel: the lane / register / LDS index maps of its three register
passes, two transposes and partner exchange reproduce numpy's FFT and power spectrum, and every LDS access
of the transform is bank-conflict free under the bank model of MI355X_MICROARCH.md (the 2-way conflicts of
the 17 power-spectrum writes are the known exception)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_index_maps_and_bank_model():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'model_fbank2048.py')],
                         capture_output=True, text=True, check=True).stdout
    fft_err = float(re.search(r'complex FFT max err (\S+)', out).group(1))
    pow_err = float(re.search(r'power max rel err (\S+)', out).group(1))
    assert fft_err < 1e-10 and pow_err < 1e-12
    conflicts = [line for line in out.splitlines() if 'bank conflict' in line]
    assert all('P write' in line and 'x2' in line for line in conflicts), conflicts
