"""Philox4x32-10 against the Random123 known-answer vectors (Salmon et al. 2011; Random123 kat_vectors, `philox4x32 10`):
the oracle's NumPy restatement, the engine's header compiled for the host (CPU) and the same function on the device (GPU).
Everything the engine and the oracle draw is a fixed map of these words (include/tinyda_amd.h, "RNG stream")."""
import ctypes as C

import numpy as np
import pytest

# counter[4], key[2] -> output[4]
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def _lib_philox(device, ctr, key):
    from tinyda_amd import _lib

    lib = _lib.load()
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    out = (C.c_uint32 * 4)()
    _lib.check(lib.tda_rng_philox(device, c, k, out))
    return tuple(int(v) for v in out)


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_oracle_philox_known_answers(ctr, key, want):
    from oracle import tinyda_oracle as orc

    got = orc.philox4x32_10(*[np.uint32(v) for v in ctr], key[0], key[1])
    assert tuple(int(np.asarray(g).reshape(-1)[0]) for g in got) == want


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_engine_header_on_host_known_answers(ctr, key, want):
    import __graft_entry__ as g

    g.build()
    assert _lib_philox(-1, ctr, key) == want


def test_oracle_stream_is_a_map_of_the_kat_words():
    """seed 0, chain 0, step 0, stream 0, block 0 is the all-zero KAT: the first two proposal normals follow from its words"""
    from oracle import tinyda_oracle as orc

    x = KAT[0][2]
    u1 = ((x[0] >> 5) * 67108864.0 + (x[1] >> 6)) / 9007199254740992.0
    u2 = ((x[2] >> 5) * 67108864.0 + (x[3] >> 6)) / 9007199254740992.0
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    want = np.array([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])
    st = orc.PhiloxStream(0)
    w = st.words(np.uint32(0), np.uint32(0), orc.STREAM_PROPOSAL, np.uint32(0))
    assert tuple(int(np.asarray(v).reshape(-1)[0]) for v in w) == x
    np.testing.assert_allclose(st.normals([0], 0, 2)[0], want, rtol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("ctr,key,want", KAT)
def test_device_philox_known_answers(ctr, key, want):
    assert _lib_philox(0, ctr, key) == want


@pytest.mark.gpu
def test_device_normals_follow_from_the_kat_words():
    """engine with seed 0, chain 0, step 0: z[0], z[1] = Box-Muller of the all-zero KAT block (RNG contract, stream 0)"""
    from tinyda_amd.engine import Engine

    x = KAT[0][2]
    u1 = ((x[0] >> 5) * 67108864.0 + (x[1] >> 6)) / 9007199254740992.0
    u2 = ((x[2] >> 5) * 67108864.0 + (x[3] >> 6)) / 9007199254740992.0
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    e = Engine(4, 2, seed=0)
    z, _ = e.rng_probe(0)
    e.close()
    np.testing.assert_allclose(z[0], [r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)], rtol=1e-13)
