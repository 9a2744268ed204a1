"""gin-config when it is installed, otherwise a minimal stand-in covering the syntax the HiD-VAE configs use
(SURVEY.md 5.6): `import a.b`, `fn.arg = <python literal>`, `%module.Enum.MEMBER`, `#` comments."""
import ast
import importlib
import inspect
import re

try:  # pragma: no cover - gin-config is not in this image
    import gin as _real_gin
except Exception:  # noqa: BLE001
    _real_gin = None

_BINDINGS = {}
_CONSTANTS = {}
_CONFIGURABLES = {}


def constants_from_enum(cls=None, module=None):
    def register(c):
        mod = module or c.__module__
        for member in c:
            _CONSTANTS[f"{mod}.{c.__name__}.{member.name}"] = member
            _CONSTANTS[f"{mod.split('.', 1)[-1]}.{c.__name__}.{member.name}"] = member  # also without the package prefix
        if _real_gin is not None:
            try:
                _real_gin.constants_from_enum(c, module=mod)
            except Exception:  # noqa: BLE001  already registered
                pass
        return c
    return register(cls) if cls is not None else register


def configurable(fn=None, **_kw):
    def wrap(f):
        name = f.__name__
        _CONFIGURABLES[name] = f
        sig = inspect.signature(f)

        def call(*args, **kwargs):
            bound = dict(_BINDINGS.get(name, {}))
            bound.update(kwargs)
            unknown = set(bound) - set(sig.parameters)
            if unknown:
                raise TypeError(f"{name}: unknown configured arguments {sorted(unknown)}")
            return f(*args, **bound)

        call.__name__ = name
        call.__wrapped__ = f
        return call
    return wrap(fn) if fn is not None else wrap


def _value(text):
    text = text.strip()
    if text.startswith("%"):
        key = text[1:]
        if key in _CONSTANTS:
            return _CONSTANTS[key]
        tail = [v for k, v in _CONSTANTS.items() if k.endswith("." + ".".join(key.split(".")[-2:]))]
        if tail:
            return tail[0]
        raise ValueError(f"unknown gin constant {text}")
    return ast.literal_eval(text)


def parse_config(text, import_aliases=None):
    import_aliases = import_aliases or {}
    for raw in text.splitlines():
        line = raw.split("#", 1)[0].strip() if not re.search(r"['\"].*#.*['\"]", raw) else raw.strip()
        if not line:
            continue
        if line.startswith("import "):
            mod = line[len("import "):].strip()
            mod = import_aliases.get(mod, mod)
            try:
                importlib.import_module(mod)
            except ImportError:
                pass  # data-ingestion modules of the reference are out of scope here
            continue
        m = re.match(r"^([A-Za-z_][\w.]*)\.([A-Za-z_]\w*)\s*=\s*(.+)$", line)
        if not m:
            raise ValueError(f"unsupported gin syntax: {raw!r}")
        scope, arg, val = m.group(1), m.group(2), m.group(3)
        _BINDINGS.setdefault(scope.split(".")[-1], {})[arg] = _value(val)


def parse_config_file(path, import_aliases=None):
    with open(path) as f:
        parse_config(f.read(), import_aliases)


def clear_config():
    _BINDINGS.clear()


def bindings(name):
    return dict(_BINDINGS.get(name, {}))
