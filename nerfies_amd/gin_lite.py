"""A small interpreter for the subset of the gin-config language the nerfies presets use.

gin-config is not installed in this image and is a third-party dependency of the reference (pinned git sha,
requirements.txt:3), so the surface the reference relies on is restated here:

  include 'file.gin'                       nested includes; looked up as written (CWD), next to the including file,
                                           then on the search path (configs/ trees use both spellings)
  name = <value>                           macro; `%name` references are LAZY -- they see the last definition,
                                           wherever it appears (gpu_vrig_paper.gin:31-32 overriding defaults.gin:19-20)
  [scope/]Class.param = <value>            binding; later statements win
  <value>                                  python literal (numbers, strings, None/True/False, tuples, lists, dicts,
                                           multi-line) with %macro and @configurable / @configurable() references
  import a.b                               accepted and ignored

API (same names as the calls in train.py:107-110 / eval.py:232-235): parse_config_files_and_bindings(config_files,
bindings, skip_unknown), configurable (class decorator: bindings become constructor defaults, explicit keyword
arguments win -- eval.py:239), external_configurable, REQUIRED, operative_config_str, clear_config, query_parameter."""
import ast
import dataclasses
import functools
import inspect
import os
import re
from typing import Any, Dict, List, Optional, Tuple

REQUIRED = type('Required', (), {'__repr__': lambda self: 'gin.REQUIRED'})()


class GinError(ValueError):
  pass


@dataclasses.dataclass(frozen=True)
class _Macro:
  name: str


@dataclasses.dataclass(frozen=True)
class _Ref:
  name: str
  call: bool


_state: Dict[str, Any] = {}
_search_paths: List[str] = []
_configurables: Dict[str, Any] = {}          # 'ModelConfig' / 'nn.softplus' / 'flax.nn.softplus' -> object
_used: Dict[Tuple[str, str], Any] = {}       # (configurable, param) -> value actually injected


def clear_config():
  _state.clear()
  _state.update(macros={}, bindings={})
  _used.clear()


clear_config()


def add_config_file_search_path(path):
  _search_paths.append(path)


# ---------------------------------------------------------------- lexer / value parser
def _strip_comment(line: str) -> str:
  quote = None
  i = 0
  while i < len(line):
    ch = line[i]
    if quote:
      if ch == '\\':
        i += 1
      elif ch == quote:
        quote = None
    elif ch in '\'"':
      quote = ch
    elif ch == '#':
      return line[:i]
    i += 1
  return line


def _depth(text: str) -> int:
  depth, quote, i = 0, None, 0
  while i < len(text):
    ch = text[i]
    if quote:
      if ch == '\\':
        i += 1
      elif ch == quote:
        quote = None
    elif ch in '\'"':
      quote = ch
    elif ch in '([{':
      depth += 1
    elif ch in ')]}':
      depth -= 1
    i += 1
  return depth


def _statements(text: str):
  """Yields (first line number, logical statement) -- a statement continues while brackets are open."""
  buf, start = '', 0
  for no, raw in enumerate(text.splitlines(), 1):
    line = _strip_comment(raw).rstrip()
    if not buf and not line.strip():
      continue
    if not buf:
      start = no
    buf = (buf + '\n' + line) if buf else line
    if _depth(buf) <= 0 and not buf.rstrip().endswith(('=', '\\')):
      yield start, buf.strip()
      buf = ''
  if buf.strip():
    raise GinError(f'line {start}: unterminated value')


_SPECIAL = re.compile(r'''("(?:\\.|[^"\\])*"|'(?:\\.|[^'\\])*')|%([A-Za-z_][\w./]*)|@([A-Za-z_][\w./]*)(\(\))?''')


def _parse_value(src: str, where: str):
  def sub(m):
    if m.group(1):
      return m.group(1)
    if m.group(2):
      return f'__gin_macro__({m.group(2)!r})'
    return f'__gin_ref__({m.group(3)!r}, {bool(m.group(4))})'
  try:
    tree = ast.parse(_SPECIAL.sub(sub, src.strip()), mode='eval')
  except SyntaxError as e:
    raise GinError(f'{where}: cannot parse value {src!r}: {e.msg}') from None

  def ev(node):
    if isinstance(node, ast.Constant):
      return node.value
    if isinstance(node, ast.Tuple):
      return tuple(ev(e) for e in node.elts)
    if isinstance(node, ast.List):
      return [ev(e) for e in node.elts]
    if isinstance(node, ast.Dict):
      return {ev(k): ev(v) for k, v in zip(node.keys, node.values)}
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
      v = ev(node.operand)
      return -v if isinstance(node.op, ast.USub) else v
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Name):
      args = [ev(a) for a in node.args]
      if node.func.id == '__gin_macro__':
        return _Macro(args[0])
      if node.func.id == '__gin_ref__':
        return _Ref(args[0], args[1])
    raise GinError(f'{where}: unsupported expression in {src!r}')
  return ev(tree.body)


_LHS = re.compile(r'^(?:(?P<scope>[\w/]+)/)?(?P<name>[A-Za-z_][\w.]*)$')


def _find_include(path: str, including_dir: Optional[str]) -> str:
  cands = [path]
  if including_dir:
    cands.append(os.path.join(including_dir, path))
    cands.append(os.path.join(including_dir, os.path.basename(path)))
  cands += [os.path.join(p, path) for p in _search_paths]
  for c in cands:
    if os.path.isfile(c):
      return c
  raise GinError(f"include '{path}': not found (tried {cands})")


def parse_config(text: str, skip_unknown: bool = False, _dir: Optional[str] = None, _origin: str = '<string>'):
  """Executes the statements of `text` against the global config."""
  for no, stmt in _statements(text):
    where = f'{_origin}:{no}'
    m = re.match(r'^include\s+(.+)$', stmt)
    if m:
      target = _parse_value(m.group(1), where)
      if not isinstance(target, str):
        raise GinError(f'{where}: include needs a quoted path')
      parse_config_file(target, skip_unknown, _dir)
      continue
    if re.match(r'^import\s+[\w.]+$', stmt):
      continue
    if '=' not in stmt:
      raise GinError(f'{where}: expected `name = value`, got {stmt!r}')
    lhs, rhs = stmt.split('=', 1)
    m = _LHS.match(lhs.strip())
    if not m:
      raise GinError(f'{where}: bad left-hand side {lhs.strip()!r}')
    value = _parse_value(rhs, where)
    name = m.group('name')
    if '.' not in name:
      if m.group('scope'):
        raise GinError(f'{where}: macros cannot be scoped')
      _state['macros'][name] = value
      continue
    target, param = name.rsplit('.', 1)
    if _lookup(target) is None and not skip_unknown:
      raise GinError(f'{where}: no configurable named {target!r} (pass skip_unknown=True to ignore)')
    key = target.rsplit('.', 1)[-1] if _lookup(target) is None else _canonical(target)
    _state['bindings'][(m.group('scope') or '', key, param)] = value


def parse_config_file(path: str, skip_unknown: bool = False, _including_dir: Optional[str] = None):
  real = _find_include(path, _including_dir)
  with open(real, 'r') as fp:
    parse_config(fp.read(), skip_unknown, os.path.dirname(os.path.abspath(real)), real)


def parse_config_files_and_bindings(config_files=None, bindings=None, skip_unknown: bool = False):
  """gin.parse_config_files_and_bindings as called in train.py:107-110: files first, then the --gin_bindings."""
  for f in ([config_files] if isinstance(config_files, str) else (config_files or [])):
    parse_config_file(f, skip_unknown)
  if bindings:
    parse_config(bindings if isinstance(bindings, str) else '\n'.join(bindings), skip_unknown, _origin='<bindings>')


# ---------------------------------------------------------------- configurables
def _lookup(name: str):
  if name in _configurables:
    return _configurables[name]
  hits = {id(v): v for k, v in _configurables.items() if k.endswith('.' + name)}
  return next(iter(hits.values())) if len(hits) == 1 else None


def _canonical(name: str) -> str:
  obj = _lookup(name)
  return getattr(obj, '__gin_name__', name.rsplit('.', 1)[-1])


def _resolve(value, trail=()):
  if isinstance(value, _Macro):
    if value.name in trail:
      raise GinError(f'macro cycle: {" -> ".join(trail + (value.name,))}')
    if value.name not in _state['macros']:
      raise GinError(f'macro %{value.name} is referenced but never defined')
    return _resolve(_state['macros'][value.name], trail + (value.name,))
  if isinstance(value, _Ref):
    obj = _lookup(value.name)
    if obj is None:
      raise GinError(f'no configurable named @{value.name}')
    return obj() if value.call else obj
  if isinstance(value, tuple):
    return tuple(_resolve(v, trail) for v in value)
  if isinstance(value, list):
    return [_resolve(v, trail) for v in value]
  if isinstance(value, dict):
    return {_resolve(k, trail): _resolve(v, trail) for k, v in value.items()}
  return value


def query_parameter(name: str):
  """'Class.param' or '%macro' -> resolved value."""
  if name.startswith('%'):
    return _resolve(_Macro(name[1:]))
  target, param = name.rsplit('.', 1)
  key = ('', _canonical(target), param)
  if key not in _state['bindings']:
    raise GinError(f'{name} is not bound')
  return _resolve(_state['bindings'][key])


def bindings_for(name: str, scope: str = '') -> Dict[str, Any]:
  out = {}
  for (sc, target, param), v in _state['bindings'].items():
    if target == name and sc in ('', scope):
      out[param] = _resolve(v)
  return out


def external_configurable(obj, name: Optional[str] = None, module: Optional[str] = None):
  """Registers an existing object (e.g. an activation) so `@module.name` / `@name` resolve to it (configs.py:27-32)."""
  name = name or getattr(obj, '__name__')
  _configurables[name] = obj
  if module:
    _configurables[f'{module}.{name}'] = obj
    _configurables[f'{module.rsplit(".", 1)[-1]}.{name}'] = obj     # '@nn.softplus' for module 'flax.nn'
  return obj


def configurable(cls=None, *, name: Optional[str] = None, module: Optional[str] = None):
  """Class decorator: constructing the class fills parameters from the bindings; explicit arguments win;
  a parameter still equal to REQUIRED afterwards is an error."""
  def wrap(c):
    cname = name or c.__name__
    init = c.__init__
    sig = inspect.signature(init)

    @functools.wraps(init)
    def __init__(self, *args, **kwargs):
      bound = sig.bind_partial(self, *args, **kwargs).arguments
      for param, value in bindings_for(cname).items():
        if param not in sig.parameters:
          raise GinError(f'{cname} has no parameter {param!r}')
        if param not in bound:
          kwargs[param] = value
          _used[(cname, param)] = value
      init(self, *args, **kwargs)
      for f in getattr(c, '__dataclass_fields__', {}):
        if getattr(self, f, None) is REQUIRED:
          raise GinError(f'{cname}.{f} is gin.REQUIRED but was neither bound nor passed')

    c.__init__ = __init__
    c.__gin_name__ = cname
    _configurables[cname] = c
    if module:
      _configurables[f'{module}.{cname}'] = c
    return c
  return wrap(cls) if cls is not None else wrap


def _literal(v) -> str:
  if callable(v) and not isinstance(v, type):
    for k, o in _configurables.items():
      if o is v and '.' in k:
        return '@' + k
    return '@' + getattr(v, '__name__', repr(v))
  if isinstance(v, dict):
    return '{' + ', '.join(f'{_literal(k)}: {_literal(x)}' for k, x in v.items()) + '}'
  if isinstance(v, tuple):
    return '(' + ', '.join(_literal(x) for x in v) + (',)' if len(v) == 1 else ')')
  if isinstance(v, list):
    return '[' + ', '.join(_literal(x) for x in v) + ']'
  return repr(v)


def operative_config_str() -> str:
  """The bindings that were actually consumed by constructed configurables, as parseable gin (train.py:138)."""
  lines, last = [], None
  for (cname, param), v in sorted(_used.items()):
    if cname != last:
      if lines:
        lines.append('')
      lines.append(f'# Parameters for {cname}:')
      last = cname
    lines.append(f'{cname}.{param} = {_literal(v)}')
  return '\n'.join(lines) + ('\n' if lines else '')
