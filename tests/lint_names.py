"""Scope-aware undefined-name check (this image has no pyflakes).  bench.py's main() and the GPU-only
branches of the package cannot execute on the CPU-only build host; a NameError there would only
surface on the MI355X box, so tests/test_lint.py runs this over every Python file of the repo."""
import ast, builtins, sys

SCOPES = (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda, ast.Module, ast.ClassDef,
          ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)


def _args(a):
    return [x.arg for x in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) +
            ([a.kwarg] if a.kwarg else [])]


def _own_bindings(scope):
    """Names bound directly in `scope` (not in nested scopes)."""
    names = set()
    if isinstance(scope, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
        names.update(_args(scope.args))
    stack = list(ast.iter_child_nodes(scope))
    while stack:
        n = stack.pop()
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            names.add(n.name)
            # decorators / defaults / bases evaluate in THIS scope but bind nothing
            continue
        if isinstance(n, ast.Lambda):
            continue
        if isinstance(n, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
            continue   # comprehension targets are local to the comprehension (walrus ignored)
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            names.add(n.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for al in n.names:
                names.add((al.asname or al.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            names.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            names.update(n.names)
        stack.extend(ast.iter_child_nodes(n))
    if isinstance(scope, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
        for gen in scope.generators:
            for t in ast.walk(gen.target):
                if isinstance(t, ast.Name):
                    names.add(t.id)
    return names


def undefined_names(path):
    tree = ast.parse(open(path).read(), path)
    base = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__class__"}
    problems = []

    def visit(scope, visible):
        own = _own_bindings(scope)
        # class-body names are not visible from nested functions, but are from the body itself
        here = visible | own
        inner_visible = visible if isinstance(scope, ast.ClassDef) else here
        stack = list(ast.iter_child_nodes(scope))
        while stack:
            n = stack.pop()
            if isinstance(n, SCOPES):
                if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef)):
                    for e in n.decorator_list + n.args.defaults + [d for d in n.args.kw_defaults if d] :
                        stack.append(e)
                elif isinstance(n, ast.ClassDef):
                    stack.extend(n.decorator_list + n.bases + [k.value for k in n.keywords])
                elif isinstance(n, ast.Lambda):
                    stack.extend(n.args.defaults + [d for d in n.args.kw_defaults if d])
                visit(n, inner_visible if not isinstance(n, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)) else here)
                continue
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in here:
                problems.append((n.lineno, n.id))
            stack.extend(ast.iter_child_nodes(n))

    visit(tree, base)
    return sorted(set(problems))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        for ln, name in undefined_names(p):
            print(f"{p}:{ln}: {name}")
