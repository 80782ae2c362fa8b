"""AddressSanitizer + UBSan over the host-only sources (FASTA reader with its threads and mmap path, table setup,
Java-compatible formatters). GPU ASan is not available on the pool, so the sanitizers run on the CPU build only."""
import os
import random
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_sources_are_clean_under_asan_ubsan(tmp_path):
    exe = tmp_path / "asan_driver"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"), "-pthread", "-o", str(exe),
           os.path.join(ROOT, "tools", "asan_driver.cpp"), os.path.join(ROOT, "plaac_amd", "csrc", "plaac_io.cpp"),
           os.path.join(ROOT, "plaac_amd", "csrc", "plaac_host.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "sanitize" in r.stderr and "not supported" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    rng = random.Random(7)
    files = {"crlf_blank.fa": ">a\nACD\n\n>b  \nQQ*\r\n>c\rNN\n", "empty.fa": "", "nohdr.fa": "no header\nACD",
             "only_hdr.fa": ">x", "gt_inside.fa": ">h > still header\nAC>D\n>second\nQ"}
    big = []
    for i in range(30000):  # large enough for the reader's parallel path (> 8 MB)
        n = rng.randint(0, 600)
        big.append(">p%d some text \n" % i)
        s = "".join(rng.choice("ACDEFGHIKLMNPQRSTVWYXBZ*-") for _ in range(n))
        nl = "\r\n" if i % 7 == 0 else "\n"
        big.extend(s[k:k + 60] + nl for k in range(0, n, 60))
        if i % 11 == 0:
            big.append("\n")
    files["big.fa"] = "".join(big)
    paths = []
    for name, text in files.items():
        p = tmp_path / name
        p.write_text(text, newline="")
        paths.append(str(p))
    paths += [os.path.join(ROOT, "tests", "golden", "kat28.fasta"), str(tmp_path / "missing.fa")]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", PLAAC_THREADS="6")
    r = subprocess.run([str(exe)] + paths, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
    assert "big.fa: status 0 nrec 30000" in r.stdout
