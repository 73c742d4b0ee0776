"""The product driven through the REFERENCE'S OWN Python binding.

`bindings/python/build_reference_binding.sh` compiles the reference's CPython modules - `stringzilla` (Str / Strs) and
`stringzillas` (python/stringzillas/*.c: DeviceScope, LevenshteinDistances, ...) - from the sources under /root/reference
and links the latter against `libstringzillas_rocm_shared.so` in place of the reference's own shim: the `stringzillas-rocm`
wheel target its setup.py:863-865 names but never defines.  Nothing of this repository's Python layer is involved: if these
tests pass, a user of the reference's Python API runs on MI355X by swapping one shared library.

The built modules live in oracle/_ref/pybinding (git-ignored, reference-derived, travel with the gpurun snapshot); the
tests skip when they are absent.
"""
import glob
import os
import random
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINDING = os.path.join(ROOT, "oracle", "_ref", "pybinding")


def _available():
    return bool(glob.glob(os.path.join(BINDING, "stringzillas*.so"))) and bool(glob.glob(os.path.join(BINDING, "stringzilla.*.so")))


if not _available() and os.path.isdir("/root/reference"):
    subprocess.run(["bash", os.path.join(ROOT, "bindings", "python", "build_reference_binding.sh")], check=False, capture_output=True)

needs_binding = pytest.mark.skipif(not _available(), reason="reference binding not built (no /root/reference here)")


@pytest.fixture(scope="module")
def modules():
    sys.path.insert(0, BINDING)
    try:
        import stringzilla as sz
        import stringzillas as szs
    finally:
        sys.path.remove(BINDING)
    return sz, szs


def _rand(rng, count, lo, hi, alphabet):
    return [bytes(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))).decode("latin-1") for _ in range(count)]


@needs_binding
def test_reference_binding_loads_against_this_library(modules):
    """No GPU needed: the reference's module initialises, resolves every `szs_*` symbol it binds from our library, and
    reports our library's errors through its own exception plumbing."""
    sz, szs = modules
    assert szs.__version__ == "5.1.2" and sz.__version__ == "5.1.2"
    for name in ("DeviceScope", "LevenshteinDistances", "LevenshteinDistancesUTF8", "NeedlemanWunschScores",
                 "SmithWatermanScores", "Fingerprints", "to_device"):
        assert hasattr(szs, name)
    import torch

    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="GPU"):
            szs.DeviceScope(gpu_device=0)
        with pytest.raises(RuntimeError, match="GPU engines only"):
            szs.LevenshteinDistances()


@needs_binding
@pytest.mark.gpu
def test_reference_binding_scores_on_the_gpu(modules, oracle):
    """The README's own call convention (python/README.md:452-560) on every engine family: Strs in unified memory (the
    binding swaps the allocator itself), symmetric calls, `out=` buffers - checked against the CPU oracle."""
    sz, szs = modules
    from stringzilla_amd import matrices

    gpu = szs.DeviceScope(gpu_device=0)
    rng = random.Random(7)

    engine = szs.LevenshteinDistances(capabilities=gpu)
    assert "cuda" in engine.__capabilities__
    distances = engine(sz.Strs(["hello", "world"]), sz.Strs(["hallo", "word"]), device=gpu)
    assert distances.shape == (2, 2) and distances.dtype == np.uint64 and distances[0, 0] == 1 and distances[1, 1] == 1

    queries, candidates = _rand(rng, 40, 0, 180, b"ACGT"), _rand(rng, 300, 0, 180, b"ACGT")
    q_bytes, c_bytes = [s.encode("latin-1") for s in queries], [s.encode("latin-1") for s in candidates]
    for costs in [(0, 1, 1, 1), (1, 3, 3, 3), (0, 2, 4, 1)]:
        engine = szs.LevenshteinDistances(match=costs[0], mismatch=costs[1], open=costs[2], extend=costs[3], capabilities=gpu)
        got = engine(sz.Strs(queries), sz.Strs(candidates), device=gpu)
        assert np.array_equal(got, oracle.levenshtein(q_bytes, c_bytes, *costs)), costs
        assert np.array_equal(engine(sz.Strs(queries), device=gpu), oracle.levenshtein(q_bytes, None, *costs)), costs
    out = np.zeros((40, 300), dtype=np.uint64)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    engine(sz.Strs(queries), sz.Strs(candidates), device=gpu, out=out)
    assert np.array_equal(out, oracle.levenshtein(q_bytes, c_bytes))

    utf8 = szs.LevenshteinDistancesUTF8(capabilities=gpu)
    texts = ["".join(rng.choice("aé中😀bñ語 ") for _ in range(rng.randint(0, 70))) for _ in range(30)]
    encoded = [t.encode() for t in texts]
    assert np.array_equal(utf8(sz.Strs(texts), sz.Strs(texts), device=gpu), oracle.levenshtein_utf8(encoded, encoded))

    proteins, others = _rand(rng, 12, 50, 400, b"ARNDCQEGHILKMFPSTWYV"), _rand(rng, 70, 50, 400, b"ARNDCQEGHILKMFPSTWYV")
    p_bytes, o_bytes = [s.encode() for s in proteins], [s.encode() for s in others]
    byte_to_class, class_costs = matrices.blosum62()
    for cls, kind, gaps in [(szs.NeedlemanWunschScores, "needleman_wunsch", (-4, -4)), (szs.NeedlemanWunschScores, "needleman_wunsch", (-11, -2)),
                            (szs.SmithWatermanScores, "smith_waterman", (-4, -1))]:
        scorer = cls(byte_to_class, class_costs, open=gaps[0], extend=gaps[1], capabilities=gpu)
        got = scorer(sz.Strs(proteins), sz.Strs(others), device=gpu)
        assert got.dtype == np.int64
        assert np.array_equal(got, getattr(oracle, kind)(p_bytes, o_bytes, byte_to_class, class_costs, *gaps)), (kind, gaps)
    # fingerprints through the reference's `szs.Fingerprints` (python/stringzillas/fingerprints.c)
    from oracle import binding

    documents = _rand(rng, 25, 0, 6000, b"ACGT") + ["", "ab"]
    for ndim, widths in [(128, None), (256, np.array([4, 8, 16], dtype=np.uint64)), (70, None)]:
        sketcher = szs.Fingerprints(ndim=ndim, window_widths=widths, seed=7, capabilities=gpu)
        hashes, counts = sketcher(sz.Strs(documents), device=gpu)
        expected = binding.oracle_fingerprints([d.encode("latin-1") for d in documents], ndim,
                                               None if widths is None else widths.tolist(), seed=7)
        assert hashes.shape == (len(documents), ndim) and hashes.dtype == np.uint32
        assert np.array_equal(hashes, expected[0]) and np.array_equal(counts, expected[1]), ndim
    # a handful of long reads: the planner routes them to the systolic tier behind the same Python call
    reads, genome = _rand(rng, 3, 3000, 3300, b"ACGT"), _rand(rng, 2, 3000, 3300, b"ACGT")
    nuc = matrices.nuc44()
    sw = szs.SmithWatermanScores(*nuc, open=-4, extend=-1, capabilities=gpu)
    expected = oracle.smith_waterman([r.encode() for r in reads], [g.encode() for g in genome], *nuc, -4, -1)
    assert np.array_equal(sw(sz.Strs(reads), sz.Strs(genome), device=gpu), expected)


WHEELS = os.path.join(ROOT, "oracle", "_ref", "wheel")


@needs_binding
@pytest.mark.gpu
def test_the_installed_wheel_scores_on_the_gpu(tmp_path, oracle):
    """The `stringzillas-rocm` WHEEL (bindings/python/setup.py; prebuilt by `__graft_entry__.build()` into oracle/_ref/wheel, which
    travels): `pip install` into a scratch target, then a FRESH interpreter that knows nothing of this repository's build tree but
    that target (and the base `stringzilla` module the wheel requires) imports `stringzillas` - the library comes out of the wheel
    through the module's RUNPATH - and scores on the GPU."""
    wheels = glob.glob(os.path.join(WHEELS, "stringzillas_rocm-*.whl"))
    if not wheels:
        pytest.skip("the wheel is not prebuilt (no /root/reference where build() ran)")
    target = tmp_path / "site"
    done = subprocess.run([sys.executable, "-m", "pip", "install", "--no-deps", "--no-index", "--target", str(target), wheels[0]], capture_output=True, text=True)
    assert done.returncode == 0, done.stdout[-1500:] + done.stderr[-1500:]
    script = """
import os, sys
import numpy as np
import torch  # (one HIP runtime per process: torch's comes first, as in stringzilla_amd/_abi.py)
import stringzilla as sz, stringzillas as szs
assert os.path.dirname(szs.__file__) == sys.argv[1], szs.__file__
loaded = [line.split()[-1] for line in open("/proc/self/maps") if "libstringzillas_rocm_shared" in line]
assert loaded and all(path.startswith(sys.argv[1]) for path in loaded), loaded  # the wheel's own copy, not the build tree's
gpu = szs.DeviceScope(gpu_device=0)
engine = szs.LevenshteinDistances(capabilities=gpu)
got = engine(sz.Strs(["kitten", "LISTEN", "ATCA", ""]), sz.Strs(["sitting", "SILENT", "CTACTCACCC", "ABC"]), device=gpu)
print("DIAGONAL", [int(got[i, i]) for i in range(4)], szs.__version__)
"""
    environment = {key: value for key, value in os.environ.items() if key != "PYTHONPATH"}
    environment["PYTHONPATH"] = os.pathsep.join([str(target), BINDING])  # BINDING: the base `stringzilla` module (Str / Strs)
    run = subprocess.run([sys.executable, "-c", script, str(target)], capture_output=True, text=True, env=environment, cwd=str(tmp_path), timeout=600)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
    assert "DIAGONAL [3, 4, 6, 3] 5.1.2" in run.stdout  # the reference's own known answers (test/similarities.cuh:613-624)
