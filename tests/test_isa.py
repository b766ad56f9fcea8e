"""Device-code check that needs no GPU: no 64-bit shift takes its amount from a wave's last allocated VGPR (an MI355X erratum found
in round 2: DESIGN.md section 9, tools/check_shift64.py, tools/probes/shift64_probe.hip, profiles/r02_shift64_probe.txt)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_shift64 as C  # noqa: E402

SNIPPET = """
_Z3badv:
	v_lshlrev_b64 v[6:7], v15, v[6:7]
	v_lshrrev_b64 v[2:3], v14, v[8:9]
	s_endpgm
	.amdhsa_kernel _Z3badv
		.amdhsa_next_free_vgpr 16
		.amdhsa_accum_offset 16
_Z4goodv:
	v_lshlrev_b64 v[6:7], v15, v[6:7]
	v_ashrrev_i64 v[2:3], v7, v[8:9]
	s_endpgm
	.amdhsa_kernel _Z4goodv
		.amdhsa_next_free_vgpr 17
		.amdhsa_accum_offset 20
_Z4agprv:
	v_lshlrev_b64 v[6:7], v23, v[6:7]
	s_endpgm
	.amdhsa_kernel _Z4agprv
		.amdhsa_next_free_vgpr 40
		.amdhsa_accum_offset 24
"""


def test_scan_flags_only_the_last_allocated_register():
    hits = C.scan(SNIPPET.splitlines())
    assert [(k, reg) for k, _, reg, _ in hits] == [("_Z3badv", 15), ("_Z4agprv", 23)]


def test_scan_flags_device_functions_conservatively():
    lines = ["_Z6helperm:", "\tv_lshlrev_b64 v[0:1], v7, v[0:1]", "\tv_lshlrev_b64 v[0:1], v6, v[0:1]", "\ts_setpc_b64 s[30:31]"]
    assert [(k, reg, arch) for k, _, reg, arch in C.scan(lines)] == [("_Z6helperm", 7, -1)]


def test_library_device_code_has_no_such_shift(capsys):
    import pytest
    if C.hipcc() is None:
        pytest.skip("no hipcc here ($HIPCC or /opt/rocm/bin/hipcc)")
    assert C.main([]) == 0, capsys.readouterr().out
