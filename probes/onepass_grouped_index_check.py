"""Index logic of attn_bwd_onepass_grouped_kernel (csrc/attention.hip), restated: a sequence padded to whole 16-query strips, 8 strips per chunk,
strip -> (sequence slot strip / sps, query strip strip % sps).  Checks for every Lq <= 128 and 0..22 sequences per image that every (sequence, query) is
staged exactly once, lies in a strip phase A walks, and gets its dQ row written exactly once by phase B (one strip per wave).  CPU only; the GPU tests cover
Lq = 30 and 8 (tests/test_kernels_gpu.py) - this is the argument for the other sequence lengths.
    python probes/onepass_grouped_index_check.py"""


def check(Lq, nseq):
    sps = (Lq + 15) >> 4
    spc = 8 // sps
    assert spc >= 1
    staged, written = {}, {}
    for c in range((nseq + spc - 1) // spc):
        nsc = min(spc, nseq - c * spc)
        nvs = nsc * sps
        for row in range(128):                                   # staging
            strip = row >> 4
            slot = strip // sps
            q = (strip - slot * sps) * 16 + (row & 15)
            if slot < nsc and q < Lq:
                key = (c * spc + slot, q)
                staged[key] = staged.get(key, 0) + 1
                assert strip < nvs
        walked = set()                                           # phase A: full pairs, then the odd strip
        for s in range(nvs >> 1):
            walked |= {2 * s, 2 * s + 1}
        if nvs & 1:
            walked.add(nvs - 1)
        assert walked == set(range(nvs))
        for wave in range(8):                                    # phase B
            if wave < nvs:
                slot = wave // sps
                for fi in range(16):
                    q = (wave - slot * sps) * 16 + fi
                    if q < Lq:
                        assert slot < nsc
                        key = (c * spc + slot, q)
                        written[key] = written.get(key, 0) + 1
    want = {(i, q) for i in range(nseq) for q in range(Lq)}
    assert set(staged) == want and all(v == 1 for v in staged.values()), (Lq, nseq)
    assert set(written) == want and all(v == 1 for v in written.values()), (Lq, nseq)


if __name__ == "__main__":
    for Lq in range(1, 129):
        for nseq in range(0, 23):
            check(Lq, nseq)
    print("every (sequence, query) staged once and written once for Lq = 1..128, 0..22 sequences per image")
