"""``python setup.py build_ext --inplace`` -- the command the reference's README documents for this directory
(/root/reference/README.md:41-44, lib/csrc/nn/setup.py).  The reference compiles its CUDA extension here; this
one builds the native libraries of clean-pvnet_amd (hipcc --offload-arch=gfx950 + the host shim) in place, next to the
package, through ``clean-pvnet_amd/_build.py`` -- the same thing ``python __graft_entry__.py`` does.  Any other
setup.py command is refused: nothing here is meant to be installed into site-packages."""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))


def main(argv):
    if "build_ext" not in argv:
        sys.exit("usage: python setup.py build_ext --inplace   (builds libpvnet_nn.so in clean-pvnet_amd/)")
    spec = importlib.util.spec_from_file_location("_pvnet_vote_build", os.path.join(ROOT, "clean-pvnet_amd", "_build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for path in (b.build_nn(verbose=True),):
        print("built", os.path.relpath(path, ROOT))


if __name__ == "__main__":
    main(sys.argv[1:])
