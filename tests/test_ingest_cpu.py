"""Host parser (ps_libsvm_parse: data/LibsvmParser.java + CTR.parseFeature + DataSource offset/step)
against the restated reference parser, on CTR-format text (label + 23 `idx:1` + 45 `idx:val`).  No GPU."""
import numpy as np
import pytest

f32 = np.float32


def make_text(rng, n, F=23, X=45, big_ids=False, blanks=True):
    lines = []
    for i in range(n):
        cols = [str(int(rng.random() < 0.3))]
        for f in range(F):
            hi = 2 ** 40 if big_ids and f == 0 else 10 ** int(rng.integers(1, 7))
            cols.append("%d:1" % int(rng.integers(0, hi)))
        for j in range(X):
            style = rng.integers(0, 5)
            v = rng.standard_normal() * 10.0 ** int(rng.integers(-3, 4))
            s = ("%.6f" % v) if style == 0 else ("%g" % v) if style == 1 else ("%.9e" % v) if style == 2 else \
                repr(float(f32(v))) if style == 3 else str(int(v))
            cols.append("%d:%s" % (F + 1 + j, s))
        lines.append(" ".join(cols))
        if blanks and i % 7 == 3:
            lines.append("")                       # blank lines are skipped (LibsvmParser returns an empty list)
    return "\n".join(lines) + ("\n" if n % 2 else "")


@pytest.mark.parametrize("threads", [1, 4])
def test_parser_matches_restated_reference(orc, threads):
    import ps_amd
    rng = np.random.default_rng(5)
    F, X, WS = 23, 45, 100000
    text = make_text(rng, 300)
    E, Xd, Y, W = orc.parse_libsvm(text, F, X, WS)
    got = ps_amd.LibsvmParser(F, X, WS, threads=threads).parse(text)
    assert got["E"].shape == (300, F)
    np.testing.assert_array_equal(got["E"], E.astype(np.int64))          # ids < 2^24: the float IS the id
    np.testing.assert_array_equal(got["X"].view(np.uint32), Xd.view(np.uint32))      # Float.parseFloat bit for bit
    np.testing.assert_array_equal(got["Y"], Y)
    np.testing.assert_array_equal(got["W"], W.astype(np.int64))


def test_ids_go_through_float_like_the_reference(orc):
    """CTR.java:57 stores the long idx in a float: ids >= 2^24 are rounded; ids_via_float=0 keeps int64."""
    import ps_amd
    text = "1 16777217:1 3000000000:1 5:0.25\n0 7:1 9:1 5:-1e-3\n"
    E, Xd, Y, W = orc.parse_libsvm(text, 2, 1, 1000)
    got = ps_amd.LibsvmParser(2, 1, 1000).parse(text)
    np.testing.assert_array_equal(got["E"], E.astype(np.int64))
    assert got["E"][0, 0] == 16777216 and got["E"][0, 1] == 3000000000 - 3000000000 % 256 + (256 if 3000000000 % 256 > 128 else 0)
    np.testing.assert_array_equal(got["W"], W.astype(np.int64))
    np.testing.assert_array_equal(got["X"], Xd)
    exact = ps_amd.LibsvmParser(2, 1, 1000, ids_via_float=False).parse(text)
    assert exact["E"][0, 0] == 16777217 and exact["E"][0, 1] == 3000000000 and exact["W"][0, 0] == 16777217 % 1000


@pytest.mark.parametrize("offset,step", [(0, 1), (0, 2), (1, 2), (2, 3)])
def test_worker_sharding_offset_step(orc, offset, step):
    import ps_amd
    rng = np.random.default_rng(9)
    text = make_text(rng, 50, F=4, X=3)
    E, Xd, Y, W = orc.parse_libsvm(text, 4, 3, 0, offset, step)
    got = ps_amd.LibsvmParser(4, 3).parse(text, offset=offset, step=step)
    np.testing.assert_array_equal(got["E"], E.astype(np.int64)); np.testing.assert_array_equal(got["X"], Xd)
    np.testing.assert_array_equal(got["Y"], Y)
    assert "W" not in got


def test_bad_lines_are_errors_not_crashes():
    import ps_amd
    from ps_amd import native as N
    p = ps_amd.LibsvmParser(2, 1)
    for bad in ("1 3:1 4:1\n", "1 3:1 x:1 5:2\n", "1 3:1 4:1 5:abc\n", "z 3:1 4:1 5:1\n", "1 3 4:1 5:1\n"):
        with pytest.raises(N.PsError):
            p.parse(bad)
    assert p.parse("")["E"].shape == (0, 2)
    assert p.parse("\n\n  \n")["Y"].shape == (0,)


@pytest.mark.parametrize("offset,step", [(0, 1), (1, 3)])
def test_the_chunked_line_index_equals_the_serial_walk(offset, step):
    """Round 6: with threads > 1 the line index is built chunk by chunk (4 MB chunks, raw line numbers from a prefix over
    per-chunk newline counts).  A text of several chunks with blank lines, CRLF ends, a line that spans a chunk boundary's
    worth of padding and no final newline parses to the same arrays as the serial walk (threads = 1)."""
    import ps_amd
    rng = np.random.default_rng(3)
    n = 90000
    lines = []
    for i in range(n):
        if i % 997 == 5:
            lines.append("   ")                                  # blank: counted as a raw line, dropped
            continue
        body = "%d %d:1 %d:1 3:%.5f" % (i & 1, rng.integers(0, 1 << 20), rng.integers(0, 1 << 20), rng.standard_normal())
        pad = " " * int(rng.integers(0, 400))                    # runs of spaces are tolerated: ~280 bytes per line, ~21 MB in all
        lines.append(body + pad + ("\r" if i % 13 == 0 else ""))
    text = "\n".join(lines).encode()                             # (no newline at the end)
    assert len(text) > (4 << 20) * 4
    a = ps_amd.LibsvmParser(2, 1, 1000, threads=1).parse(text, offset=offset, step=step)
    b = ps_amd.LibsvmParser(2, 1, 1000, threads=6).parse(text, offset=offset, step=step)
    assert a["Y"].size == b["Y"].size > n // step - n // 900 - 2
    for k in ("E", "X", "Y", "W"):
        np.testing.assert_array_equal(a[k], b[k])
