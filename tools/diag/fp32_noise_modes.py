"""fp32_noise.py under both f32 multiplication modes (split-bf16 default, exact-f32 MFMA): same box, same inputs"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for mode in ("1", "0"):
    print(f"==== spcl_conv_set_f32_split({mode})", flush=True)
    code = ("import sys, runpy; sys.argv = ['fp32_noise.py']; import spcl_amd; from spcl_amd import native as n; "
            f"n.call('spcl_conv_set_f32_split', {mode}); runpy.run_path(r'{here}/fp32_noise.py', run_name='__main__')")
    subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(here)))
