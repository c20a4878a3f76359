"""GPU parity of the HIP SDF raster (through the C ABI, include/vgsdf.h) against the CPU
oracle on identical segment lists.  Bit-exact u8 is the bar."""
import numpy as np
import pytest

from conftest import FIRA, NOTO, noto_files

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(vg):
    c = vg.SdfContext(0)
    yield c
    c.close()


def oracle_jobs(font, cps):
    jobs = []
    for cp in cps:
        r = font.prepare_glyph(int(cp))
        if r is None:
            continue
        info, segs = r
        if info.has_bitmap:
            jobs.append((cp, info, segs))
    return jobs


def check_against_oracle(oracle, vg, ctx, jobs, mode):
    batch = vg.make_batch((segs, info.x0, info.y0, info.w, info.h) for _, info, segs in jobs)
    out = ctx.render_batch(batch)
    bad = []
    for g, (cp, info, segs) in enumerate(jobs):
        want = oracle.sdf_render(segs, info.x0, info.y0, info.w, info.h, mode)
        got = batch.bitmap(out, g)
        if not np.array_equal(want, got):
            bad.append((cp, int(np.count_nonzero(want != got))))
    assert not bad, f"{len(bad)} glyphs differ, first: {bad[:5]}"


def test_square_kat(oracle, vg, ctx):
    segs = np.array([[1, 2, 5, 2], [5, 2, 5, 6], [5, 6, 1, 6], [1, 6, 1, 2]], dtype=np.float64)
    batch = vg.make_batch([(segs, -2, -1, 10, 10)])
    out = batch.bitmap(ctx.render_batch(batch), 0)
    assert np.array_equal(out, oracle.sdf_render(segs, -2, -1, 10, 10, oracle.PRECISE))


def test_fira_block0_bit_exact(oracle, vg, ctx, fira_oracle):
    jobs = oracle_jobs(fira_oracle, range(256))
    assert len(jobs) == 189
    check_against_oracle(oracle, vg, ctx, jobs, oracle.PRECISE)


def test_fira_all_bit_exact(oracle, vg, ctx, fira_oracle):
    jobs = oracle_jobs(fira_oracle, fira_oracle.codepoints())
    assert len(jobs) == 1679
    check_against_oracle(oracle, vg, ctx, jobs, oracle.BRUTE)


def test_noto_regular_bit_exact(oracle, vg, ctx, noto_oracle):
    cps = noto_oracle.codepoints()
    jobs = oracle_jobs(noto_oracle, cps[cps <= 0xFFFF])
    assert len(jobs) == 2973
    check_against_oracle(oracle, vg, ctx, jobs, oracle.BRUTE)
