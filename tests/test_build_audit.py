"""The code-object audit's check on hand-issued LDS reads (animate_anything_amd/build.py: pending_lds_read_violation): between an
`asm volatile` ds_read and the counted s_waitcnt that retires it nothing may name its destination registers."""
from animate_anything_amd.build import pending_lds_read_violation as check


def test_sound_sequences_pass():
    body = """
        ds_read_b128 v[4:7], v1 offset:4096
        ds_read_b128 v[8:11], v1 offset:8192
        v_add_f32_e32 v20, v21, v22
        s_waitcnt lgkmcnt(1)
        v_mfma_f32_32x32x16_f16 a[0:15], v[4:7], v[30:33], a[0:15]
        ds_read_b64_tr_b16 v[4:5], v2
        s_waitcnt vmcnt(3) lgkmcnt(0)
        v_mov_b32_e32 v40, v9
        v_mov_b32_e32 v41, v4
    """.split("\n")
    assert check(body) is None


def test_use_before_the_wait_is_reported():
    body = """
        ds_read_b128 v[4:7], v1
        v_mov_b32_e32 v40, v5
        s_waitcnt lgkmcnt(0)
    """.split("\n")
    assert "v_mov_b32_e32 v40, v5" in check(body)


def test_counted_wait_retires_only_the_oldest():
    body = """
        ds_read_b128 v[4:7], v1
        ds_read_b128 v[8:11], v1 offset:16
        s_waitcnt lgkmcnt(1)
        v_mfma_f32_32x32x16_f16 a[0:15], v[8:11], v[30:33], a[0:15]
    """.split("\n")
    assert "v_mfma" in check(body)


def test_a_vmcnt_wait_retires_nothing_and_pending_addresses_are_reported():
    body = """
        ds_read_b32 v3, v1
        s_waitcnt vmcnt(0)
        ds_read_b128 v[8:11], v3
    """.split("\n")
    assert "LDS access uses a register" in check(body)
