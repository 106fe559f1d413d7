"""`import derl` drop-in: the import name of mknbv/derl bound to the MI355X-native package.

``derl`` and every ``derl.<submodule>`` ARE the ``derl_amd`` modules (same module objects, no
second copy of any class), so ``import derl; derl.PPOFactory``, ``derl.env.make(...)`` and
``from derl.runners import GAE`` work on code written against the reference
(/root/reference/derl/__init__.py lists the names; the on-policy part of them exists here)."""
import importlib
import importlib.abc
import importlib.util
import sys

import derl_amd

_ALIAS, _REAL = __name__, derl_amd.__name__


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
  """Resolves ``derl.x.y`` to the already-imported (or importable) ``derl_amd.x.y``."""

  def find_spec(self, fullname, path=None, target=None):
    if not fullname.startswith(_ALIAS + "."):
      return None
    return importlib.util.spec_from_loader(fullname, self)

  def create_module(self, spec):
    return importlib.import_module(_REAL + spec.name[len(_ALIAS):])

  def exec_module(self, module):
    """The real module is already initialised."""


sys.meta_path.insert(0, _AliasFinder())
for _name, _module in list(sys.modules.items()):
  if _name == _REAL or _name.startswith(_REAL + "."):
    sys.modules[_ALIAS + _name[len(_REAL):]] = _module
