"""CPU tier: the receiver-side state machine of the C++ host mirror (SURVEY.md section 8a row A10:
ReceptionMode / ReceptionEvent.execute / Transciever.setReceiving, clearReceiving, getRSSI,
getReceivingState) -- compiled with g++, linked against the C ABI library, no GPU call."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reception_state_machine(rsa, tmp_path):
    lib = os.path.dirname(rsa.library_path())
    exe = os.path.join(str(tmp_path), "host_state_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "host_state_test.cpp"), "-L" + lib, "-lradiomedium_hip",
                           "-Wl,-rpath," + lib])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
