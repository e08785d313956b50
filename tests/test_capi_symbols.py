"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/molahip.h declares, the ctypes structs match the C layouts, and calls fail loudly (no CPU
fallback) when no HIP device is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

from mola_lidar_odometry_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "molahip.h")


def declared_functions():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"MH_API\s+[\w\s\*]+?\b(mh_\w+)\s*\(", src)))


def test_header_declares_expected_surface():
    fns = declared_functions()
    for must in ("mh_ctx_create", "mh_map_build", "mh_scan_create", "mh_nn_search", "mh_gn_solve", "mh_covariance",
                 "mh_icp_align", "mh_icp_align_batch", "mh_last_error_string"):
        assert must in fns
    assert len(fns) >= 23


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    for fn in declared_functions():
        assert hasattr(L, fn), f"{fn} declared in molahip.h but not exported by libmolahip.so"
    # and the python binding covers every declared function
    assert sorted(capi._SIGNATURES) == declared_functions()


def test_no_cxx_or_torch_types_cross_the_boundary():
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH], text=True)
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert exported and all(s.startswith("mh_") for s in exported), exported
    decls = re.findall(r"MH_API[^;]+;", open(HEADER).read())
    for d in decls:  # plain C types only in the signatures
        assert not re.search(r"torch|at::|std::|Tensor|&", d), d


def test_struct_layouts_match_c(tmp_path):
    """Compile a tiny C program that prints sizeof/offsetof and compare with the ctypes mirrors."""
    prog = tmp_path / "sz.c"
    prog.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "molahip.h"
int main(void){
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(mh_map_params), sizeof(mh_map_info), sizeof(mh_pairs_out),
    sizeof(mh_match_info), sizeof(mh_pairs_pt2pt), sizeof(mh_pairs_pt2pl), sizeof(mh_prior), sizeof(mh_gn_params),
    sizeof(mh_gn_step), sizeof(mh_icp_params), sizeof(mh_icp_result));
  printf("%zu %zu %zu %zu\n", sizeof(mh_icp_iter), offsetof(mh_icp_params, gn), offsetof(mh_icp_params, hook_checkpoint),
    offsetof(mh_icp_result, match_kernel_ms));
  printf("%zu %zu %zu\n", sizeof(mh_preprocess_params), offsetof(mh_preprocess_params, bbox_mode),
    offsetof(mh_preprocess_params, time_offset));
  return 0; }''')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    a, b, c = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    assert [int(v) for v in c.split()] == [C.sizeof(capi.PreprocessParams), capi.PreprocessParams.bbox_mode.offset,
                                           capi.PreprocessParams.time_offset.offset]
    sizes = [int(v) for v in a.split()]
    mirrors = [capi.MapParams, capi.MapInfo, capi.PairsOut, capi.MatchInfo, capi.PairsPt2Pt, capi.PairsPt2Pl, capi.Prior,
               capi.GNParamsC, capi.GNStep, capi.ICPParamsC, capi.ICPResult]
    assert sizes == [C.sizeof(m) for m in mirrors]
    s_iter, off_gn, off_chk, off_ms = [int(v) for v in b.split()]
    assert s_iter == C.sizeof(capi.ICPIter)
    assert off_gn == capi.ICPParamsC.gn.offset and off_chk == capi.ICPParamsC.hook_checkpoint.offset
    assert off_ms == capi.ICPResult.match_kernel_ms.offset


def test_enums_shared_with_oracle(oracle):
    assert [capi.KERNEL_NONE, capi.KERNEL_GM_C4, capi.KERNEL_GM_KISS, capi.KERNEL_GM_BARRON, capi.KERNEL_CAUCHY,
            capi.KERNEL_GM_C2] == [oracle.KERNEL_NONE, oracle.KERNEL_GM_C4, oracle.KERNEL_GM_KISS,
                                   oracle.KERNEL_GM_BARRON, oracle.KERNEL_CAUCHY, oracle.KERNEL_GM_C2]
    assert capi.TERM_NAMES == oracle.TERM_NAMES


def test_version_and_status_strings():
    L = capi.lib()
    a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert L.mh_version(C.byref(a), C.byref(b), C.byref(c)) == 0
    assert (a.value, b.value) == (0, 1)
    assert L.mh_status_string(0) == b"MH_OK" and L.mh_status_string(5) == b"MH_ERR_NO_DEVICE"


def test_fails_loudly_without_a_device():
    """The product path has no CPU fallback: on a box without a GPU, context creation must raise."""
    if capi.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(capi.MolahipError) as e:
        capi.Context(0)
    assert e.value.status == 5 and "no CPU fallback" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mola_lidar_odometry_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_c" not in txt and "icp_oracle" not in txt and "libicp_oracle" not in txt, f
