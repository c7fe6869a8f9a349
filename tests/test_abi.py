"""CPU tests of the drop-in boundary: libce_hip.so loads without a GPU, exports every
symbol include/ce_api.h declares, and the ctypes mirror matches the header's structs."""
import ctypes
import os
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "ce_api.h").read_text()


def declared_functions():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(ce_[a-z0-9_]+)\s*\(", body)))


def test_build_and_load():
    import __graft_entry__ as g
    g.build()
    import cachedembedding_amd as ce
    assert ce.LIB_PATH.exists()
    assert ce._lib.lib.ce_version() == 6


def test_library_on_disk_is_the_one_the_sources_describe():
    """build() decides "up to date" from the stamp that lies with the objects (untracked, like the library), not from the
    tracked csrc/.build_stamp -- a checkout can put an old tracked stamp back beside a newer library."""
    from cachedembedding_amd import build as b
    b.build()
    tracked = b.PKG / "csrc" / ".build_stamp"
    own = b.PKG / "csrc" / "build" / "lib.stamp"
    assert own.read_text().strip() == b._digest() == tracked.read_text().strip()
    before = b.LIB.stat().st_mtime_ns
    saved = own.read_text()
    try:
        own.write_text("0" * 64)                 # "the library was built from something else"
        b.build()
        assert b.LIB.stat().st_mtime_ns > before and own.read_text().strip() == b._digest()
    finally:
        if own.read_text().strip() != b._digest():
            own.write_text(saved)


def test_every_declared_symbol_is_exported_and_bound():
    from cachedembedding_amd import _lib
    names = declared_functions()
    assert len(names) >= 25
    out = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (ce_[a-z0-9_]+)", out))
    assert set(names) <= exported, sorted(set(names) - exported)
    assert set(names) == set(_lib.SIGNATURES), (set(names) ^ set(_lib.SIGNATURES))
    for n in names:
        assert getattr(_lib.lib, n) is not None


def test_product_library_has_no_test_hooks_and_few_switches():
    """the fault / delay injection the worker tests need is compiled only into libce_hip_testhooks.so
    (-DCE_TEST_HOOKS); the product library carries neither the switches' names nor more than 20 environment switches
    in all; both libraries export the same C ABI."""
    from cachedembedding_amd import build as b
    lib, hooks = b.build(), b.build_test_hooks()
    blob, hblob = lib.read_bytes(), hooks.read_bytes()
    for name in (b"CE_WORKER_FAIL_IN_JOB", b"CE_WORKER_OUT_DELAY_US"):
        assert name not in blob, name
        assert name in hblob, name
    for gone in (b"CE_MARK_DEDUPE", b"CE_EARLY_MAPS", b"CE_SPLIT_AFTER_EMIT", b"CE_FWD_STORE", b"CE_BWD_BLOCKS_PER_CU",
                 b"CE_FWDK_BLOCKS_PER_CU", b"CE_PRESORT_STAGED", b"CE_DEDUPE_IT", b"CE_BWD_DEBUG"):
        assert gone not in blob, gone
    switches = set()
    for src in sorted((b.PKG / "csrc").glob("*.*")):
        if src.suffix in (".hip", ".h", ".cpp"):
            switches |= set(re.findall(r'getenv\("(CE_[A-Z0-9_]+)"\)', src.read_text()))
    assert len(switches) <= 20, sorted(switches)

    def exports(path):
        out = subprocess.run(["nm", "-D", "--defined-only", str(path)], capture_output=True, text=True).stdout
        return set(re.findall(r" T (ce_[a-z0-9_]+)", out))
    assert exports(lib) == exports(hooks)


def _struct_fields(name):
    m = re.search(r"typedef struct %s \{(.*?)\} %s_t;" % (name, name), HEADER, flags=re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    return [re.split(r"[\s\*]+", d.strip())[-1] for d in body.split(";") if d.strip()]


def test_struct_mirrors_match_header():
    from cachedembedding_amd import _lib
    assert [f for f, _ in _lib.CeCacheConfig._fields_] == _struct_fields("ce_cache_config")
    assert [f for f, _ in _lib.CeCallStats._fields_] == _struct_fields("ce_call_stats")
    assert ctypes.sizeof(_lib.CeCallStats) == 64
    assert ctypes.sizeof(_lib.CeCacheConfig) == 112


def test_constants_match_header():
    from cachedembedding_amd import _lib
    for name, val in re.findall(r"#define (CE_[A-Z_]+) (\d+)", HEADER):
        if hasattr(_lib, name):
            assert getattr(_lib, name) == int(val), name


def test_argument_validation_without_gpu():
    """Pure host-side checks run before any HIP call, so they are testable here."""
    from cachedembedding_amd import _lib
    lib = _lib.lib
    assert lib.ce_cache_workspace_bytes(0, 0, 0, 8) == 0
    assert lib.ce_cache_workspace_bytes(1000, 50, 64, 8) > 0
    assert lib.ce_bucketize_workspace(4096, 8) > 0
    rc = lib.ce_bag_forward(None, 10, 0, None, 0, None, 0, 4, 1, None, 0, 0, None, None)
    assert rc == _lib.CE_ERR_INVALID and "null pointer" in _lib.last_error()
    rc = lib.ce_cache_create(None, None, None)
    assert rc == _lib.CE_ERR_INVALID


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import cachedembedding_amd as ce
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ce.CachedEmbeddingBag(100, 8, cache_ratio=0.1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ce.embedding_bag(torch.zeros(4, dtype=torch.long), torch.zeros(10, 8), torch.arange(5), mode="sum",
                         include_last_offset=True)


def test_product_never_imports_oracle():
    for p in (ROOT / "cachedembedding_amd").rglob("*.py"):
        txt = p.read_text()
        assert "oracle" not in re.sub(r'""".*?"""', "", txt, flags=re.S).replace("# oracle", ""), p


def test_transport_choice_and_tracing_helpers():
    """host logic that needs no GPU: the size rule of the automatic transport choice, the phase ranges"""
    from cachedembedding_amd.pipeline import AUTO_WORKER_MIN_IDS, pick_transport
    assert pick_transport("auto", AUTO_WORKER_MIN_IDS) == "worker"
    assert pick_transport("auto", 16384 * 26) == "worker"              # one Criteo batch (prefetch_num = 1)
    assert pick_transport("auto", 2048 * 22) == "zerocopy"             # the micro-benchmark shape
    assert pick_transport("auto", 8 * 16384 * 26) == "worker"          # the bench window
    assert pick_transport("staged", 10) == "staged" and pick_transport(None, 10 ** 9) is None
    from cachedembedding_amd.tracing import phase
    with phase("prefetch cache"):
        with phase("forward pass"):
            pass
    import cachedembedding_amd as ce
    assert 1 <= ce._lib.lib.ce_cpu_budget() <= (os.cpu_count() or 1)
    assert ce._lib.lib.ce_cache_phase_count() == 6
    assert ce._lib.lib.ce_cache_phase_name(4) == b"admit_swap"
