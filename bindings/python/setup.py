"""stringzillas-rocm - the wheel target the reference declares (`sz_target == "stringzillas-rocm"`, /root/reference/setup.py:863-865)
and never defines: the reference's OWN CPython binding sources (`python/stringzillas/*.c`, compiled unmodified from a StringZilla
checkout - nothing of it is copied into this repository) over this repository's `libstringzillas_rocm_shared.so`.

    STRINGZILLA_SOURCE=/path/to/StringZilla  python -m pip wheel bindings/python --no-build-isolation --no-deps -w dist/
    pip install dist/stringzillas_rocm-*.whl stringzilla==<same version>       # then: import stringzillas

What the wheel holds: the top-level extension module `stringzillas` (the name every StringZillas wheel installs, so that
`import stringzillas` is the same line whichever backend is installed) and, beside it, `stringzillas_rocm_libs/
libstringzillas_rocm_shared.so.5` - found through the module's RUNPATH `$ORIGIN/stringzillas_rocm_libs`, like a wheel that
auditwheel repaired.  The HIP library is built by `stringzilla_amd/csrc/Makefile` (hipcc, gfx950) unless STRINGZILLAS_ROCM_LIBDIR
names a directory that already holds it.  Like the reference's GPU wheels it depends on the base `stringzilla` wheel of the same
version (setup.py:869-873) for `Str` / `Strs`.
"""
import glob
import os
import re
import shutil
import subprocess

from setuptools import Extension, setup
from setuptools.command.build_ext import build_ext

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("STRINGZILLA_SOURCE", os.environ.get("REFERENCE", "/root/reference"))
LIBDIR = os.environ.get("STRINGZILLAS_ROCM_LIBDIR", os.path.join(ROOT, "stringzilla_amd", "lib"))
LIBRARY = "libstringzillas_rocm_shared.so"
SONAME = LIBRARY + ".5"


def version():
    """The version of the C-ABI this library implements (include/stringzillas/stringzillas.h restates the reference's 5.1.2)."""
    try:
        with open(os.path.join(REFERENCE, "VERSION")) as handle:
            return handle.read().strip()
    except OSError:
        with open(os.path.join(ROOT, "CMakeLists.txt")) as handle:
            return re.search(r"project\(stringzillas_rocm VERSION ([0-9.]+)", handle.read()).group(1)


class build_over_the_rocm_library(build_ext):
    def run(self):
        if not os.path.isdir(os.path.join(REFERENCE, "python", "stringzillas")):
            raise SystemExit(f"STRINGZILLA_SOURCE={REFERENCE} holds no python/stringzillas: point it at a StringZilla checkout")
        if not os.path.exists(os.path.join(LIBDIR, LIBRARY)):  # hipcc cross-compiles gfx950 without a GPU
            subprocess.run(["make", "-C", os.path.join(ROOT, "stringzilla_amd", "csrc"), "-j8", "ARCH=gfx950"], check=True)
        super().run()
        # the library travels inside the wheel, under its SONAME, where the module's RUNPATH looks for it
        target = os.path.join(self.build_lib, "stringzillas_rocm_libs")
        os.makedirs(target, exist_ok=True)
        shutil.copy2(os.path.join(LIBDIR, LIBRARY), os.path.join(target, SONAME))
        with open(os.path.join(target, "__init__.py"), "w") as handle:
            handle.write('"""Holds libstringzillas_rocm_shared.so.5 for the `stringzillas` extension module (RUNPATH $ORIGIN/stringzillas_rocm_libs)."""\n')


import numpy  # noqa: E402  (the binding's NumPy views: python/stringzillas/similarities.c)

macros = [("SZ_DYNAMIC_DISPATCH", "1"), ("SZ_USE_CUDA", "1"), ("FU_WITH_TOPOLOGY", "0"),  # the reference's GPU target (setup.py:837): the
          # binding then allocates through `szs_unified_alloc` and feeds `sz_cap_cuda_k` - the bit this library reports and requires
          ("SZ_IS_BIG_ENDIAN_", "0"), ("SZ_IS_64BIT_X86_", "1"), ("SZ_IS_64BIT_ARM_", "0"), ("SZ_USE_WESTMERE", "1"), ("SZ_USE_GOLDMONT", "1"),
          ("SZ_USE_HASWELL", "1"), ("SZ_USE_SKYLAKE", "1"), ("SZ_USE_ICELAKE", "1"), ("SZ_USE_NEON", "0"), ("SZ_USE_NEONAES", "0"),
          ("SZ_USE_NEONSHA", "0"), ("SZ_USE_SVE", "0"), ("SZ_USE_SVE2", "0"), ("SZ_USE_SVE2AES", "0"), ("_GNU_SOURCE", "1")]
sources = sorted(glob.glob(os.path.join(REFERENCE, "python", "stringzillas", "*.c")))

setup(
    name="stringzillas-rocm",
    version=version(),
    description="Search, hash, sort, fingerprint, and fuzzy-match strings faster via SWAR, SIMD, and ROCm on AMD GPUs",  # setup.py:864-866
    license="Apache-2.0",
    python_requires=">=3.8",
    install_requires=[f"stringzilla=={version()}"],  # versions in lockstep, like the reference's multi-backend wheels
    ext_modules=[Extension(
        "stringzillas", sources=sources, define_macros=macros, language="c",
        include_dirs=[os.path.join(REFERENCE, "include"), os.path.join(REFERENCE, "c", "stringzillas"), numpy.get_include()],
        libraries=["stringzillas_rocm_shared"], library_dirs=[LIBDIR],
        extra_compile_args=["-std=c99", "-O2", "-w"], extra_link_args=["-Wl,-rpath,$ORIGIN/stringzillas_rocm_libs"])],
    cmdclass={"build_ext": build_over_the_rocm_library},
    zip_safe=False,
)
