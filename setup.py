"""Installs the MI355X-native on-policy engine as a drop-in for mknbv/derl: the packages
``derl_amd`` (product) and ``derl`` (import-name alias), the prebuilt / freshly built
``libderl_amd.so`` and the ``derl`` launcher (reference: setup.py:11-12, ``packages=["derl"]``,
``scripts=["derl/scripts/derl"]``).  ``pip install --no-build-isolation .`` compiles the HIP
sources for gfx950 with hipcc first (``derl_amd/build.py``)."""
import importlib.util
import os

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class BuildWithNativeLibrary(build_py):
  """Compiles libderl_amd.so in-tree (stale objects only) before the package files are copied."""

  def run(self):
    spec = importlib.util.spec_from_file_location("_derl_amd_build", os.path.join(ROOT, "derl_amd", "build.py"))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    module.build_library(verbose=True)
    super().run()


setup(
    name="derl-amd",
    version="0.2.0",
    description="MI355X-native rollout + PPO/A2C update engine behind mknbv/derl's API",
    license="MIT",
    python_requires=">=3.8",
    packages=find_packages(include=["derl_amd", "derl_amd.*", "derl"]),
    package_data={"derl_amd": ["libderl_amd.so", "csrc/*.hip", "csrc/*.hpp"]},
    scripts=["derl_amd/scripts/derl"],
    install_requires=["numpy>=1.16.4", "torch>=2.0"],
    extras_require={"logging": ["tensorboard>=1.15", "tqdm"]},
    cmdclass={"build_py": BuildWithNativeLibrary},
)
