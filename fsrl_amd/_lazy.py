"""Table-driven lazy exports (PEP 562) for the sub-packages: `from fsrl_amd.policy import CVPO` imports only
fsrl_amd/policy/cvpo.py (and what it needs), not every policy module and its torch dependencies."""
import importlib
from typing import Dict


def install(package: str, namespace: dict, table: Dict[str, str]) -> None:
    """table: exported name -> submodule (relative to `package`) that defines it."""
    def __getattr__(name):
        try:
            mod = importlib.import_module(f"{package}.{table[name]}")
        except KeyError:
            raise AttributeError(f"module {package!r} has no attribute {name!r}") from None
        value = getattr(mod, name)
        namespace[name] = value          # cache: later lookups bypass __getattr__
        return value

    namespace["__getattr__"] = __getattr__
    namespace["__all__"] = list(table)
    namespace["__dir__"] = lambda: sorted(set(namespace) | set(table))
