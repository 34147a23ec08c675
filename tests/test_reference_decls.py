"""Build-container check of the drop-in boundary: shim/*.cc is type-checked against declaration-only mocks
(tests/shim_compile/mock/reference_decls.hpp) because the reference's headers need OpenCV / Eigen / Sophus.  A mock that
drifted from the reference would still compile -- so every member the mock declares for the reference's classes is
looked up in the reference's own header, by name and normalised signature.  /root/reference does not exist on the GPU
box: the test skips without it, nothing of the reference travels."""
import os
import re

import pytest

REF = os.environ.get("VIEO_REFERENCE_ROOT", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include")), reason="no reference checkout here")

# mock class -> the reference headers that declare it (a class may inherit members: bases listed too)
HEADERS = {
    "FrameBase": ["include/FrameBase.h"],
    "Frame": ["include/Frame.h", "include/FrameBase.h"],
    "KeyFrame": ["include/KeyFrame.h", "include/FrameBase.h"],
    "MapPoint": ["include/MapPoint.h"],
    "Map": ["include/Map.h"],
    "ORBmatcher": ["include/ORBmatcher.h"],
    "Optimizer": ["include/Optimizer.h"],
    "Tracking": ["include/Tracking.h"],
    "LocalMapping": ["include/LocalMapping.h"],
    "IMUInitialization": ["src/Odom/IMUInitialization.h"],
    "IMUDataBase": ["src/Odom/OdomData.h"],
    "OdomPreIntegratorBase": ["src/Odom/OdomPreIntegrator.h"],
    "IMUPreIntegratorBase": ["src/Odom/OdomPreIntegrator.h"],
    "KB8Camera": ["common/camera_models/camera_kb8.h"],
}
# members the mock states in a reduced form on purpose, with the reason
KNOWN = {
    ("KeyFrame", "mbPrior"): "const bool in the reference too; the mock gives it an initialiser so that the class is constructible",
}


# member functions the reference generates with a macro: (class, name) -> the macro invocation that must be in its header
MACRO_MADE = {
    ("IMUInitialization", "GetVINSInited"): r"CREATOR_VAR_MULTITHREADS\(\s*VINSInited\s*,\s*bool\b",  # common/macro_creator.h:12-28: bool GetVINSInited(void)
}


def _strip(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    return text


def _class_body(text, name):
    """the text between the braces of `class name` / `struct name` (first definition), nested braces included"""
    m = re.search(r"\b(?:class|struct)\s+(?:[A-Z_]+\s+)?%s\b[^;{]*\{" % re.escape(name), text)  # (an export macro may sit between)
    if not m:
        return None
    i, depth = m.end(), 1
    while i < len(text) and depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i - 1]


def _norm_type(t):
    t = re.sub(r"\b(?:virtual|static|inline|explicit|override|typename|struct|class|VIEO_SLAM::|std::|Eigen::)\b", " ", t)
    t = t.replace("VIEO_SLAM::", "").replace("std::", "").replace("Eigen::", "")
    t = re.sub(r"\bconst\b", " ", t)            # const placement / top-level const is not what a drift looks like
    t = re.sub(r"\bunsigned long int\b", "unsigned long", t)
    t = re.sub(r"\blong unsigned int\b", "unsigned long", t)
    t = re.sub(r"\(\s*void\s*\)", "()", t)
    t = re.sub(r"\s+", "", t)
    return t


def _split_params(p):
    out, depth, cur = [], 0, ""
    for ch in p:
        if ch in "<([{":
            depth += 1
        elif ch in ">)]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _param_type(p):
    p = re.sub(r"=.*$", "", p.strip(), flags=re.S)          # default argument
    m = re.match(r"^(.*?)([A-Za-z_]\w*)?\s*(\[\s*\d*\s*\])?$", p.strip(), flags=re.S)
    ty, name = m.group(1), m.group(2)
    if name and (not ty.strip() or re.search(r"[\w>]\s*$", ty) is None and ty.strip()[-1] not in "&*>"):
        ty = p.strip()                                       # unnamed parameter: all of it is the type
    if name and name in ("int", "float", "double", "bool", "char", "size_t", "void", "long", "unsigned"):
        ty = p.strip()
    return _norm_type(ty)


def _functions(body):
    """name -> list of (normalised return type, tuple of normalised parameter types)"""
    flat = re.sub(r"\s+", " ", body)
    # drop inline function bodies so that their statements are not read as declarations
    out = {}
    for m in re.finditer(r"([\w:<>,&*\s~]+?)\b(~?[A-Za-z_]\w*)\s*\(([^()]*(?:\([^()]*\)[^()]*)*)\)\s*(?:const)?\s*(?:override)?\s*(?:=\s*0)?\s*[;{]", flat):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        if name in ("if", "for", "while", "switch", "return", "sizeof", "assert", "static_cast", "dynamic_cast"):
            continue
        ret = ret.split(";")[-1].split("}")[-1].split("{")[-1]
        ret = re.sub(r"\b(?:public|protected|private)\s*:", " ", ret)
        ptypes = tuple(_param_type(p) for p in _split_params(params) if p.strip() and p.strip() != "void")
        out.setdefault(name, []).append((_norm_type(ret), ptypes))
    return out


def _data_members(body):
    """name -> normalised type, for `T a, b;` declarations at class level (nested struct bodies removed by the caller)"""
    flat = re.sub(r"\s+", " ", body)
    out = {}
    for stmt in flat.split(";"):
        stmt = re.sub(r"\b(?:public|protected|private)\s*:", " ", stmt).strip()
        if not stmt or stmt.startswith(("typedef", "using", "friend", "enum", "template")):
            continue
        if "(" in stmt and re.search(r"\s=\s", stmt.split("(")[0]):
            stmt = re.sub(r"\s=\s.*$", "", stmt)           # `T name = T(args)`: an initialiser with parentheses
        if "(" in stmt:
            continue
        stmt = re.sub(r"=\s*[^,]+", "", stmt).strip()        # initialisers
        m = re.match(r"^(.*?[\w>&*\]])\s+((?:[*&]?\s*[A-Za-z_]\w*(?:\s*\[[^\]]*\])?\s*,\s*)*[*&]?\s*[A-Za-z_]\w*(?:\s*\[[^\]]*\])?)$", stmt)
        if not m:
            continue
        ty = m.group(1)
        for nm in m.group(2).split(","):
            nm = nm.strip()
            arr = re.search(r"\[[^\]]*\]", nm)
            nm2 = re.sub(r"\[[^\]]*\]", "", nm).strip("*& ")
            out[nm2] = _norm_type(ty + (arr.group(0) if arr else "") + ("*" if nm.startswith("*") else ""))
    return out


def _without_nested(body):
    """class-level text with the bodies of nested structs / inline functions blanked"""
    out, depth = "", 0
    for ch in body:
        if ch == "{":
            depth += 1
            out += "{" if depth == 1 else ""
            continue
        if ch == "}":
            depth -= 1
            out += "};" if depth == 0 else ""
            continue
        if depth == 0:
            out += ch
    return out


def _mock():
    with open(os.path.join(ROOT, "tests", "shim_compile", "mock", "reference_decls.hpp")) as f:
        return _strip(f.read())


def _ref_text(paths):
    t = ""
    for p in paths:
        with open(os.path.join(REF, p), errors="replace") as f:
            t += "\n" + _strip(f.read())
    return t


@pytest.mark.parametrize("cls", sorted(HEADERS))
def test_mock_members_exist_in_the_reference_header(cls):
    mock_body = _class_body(_mock(), cls)
    assert mock_body, cls
    ref = _ref_text(HEADERS[cls])
    ref_bodies = [b for b in (_class_body(ref, c) for c in ([cls] + (["FrameBase"] if cls in ("Frame", "KeyFrame") else []))) if b]
    assert ref_bodies, "class %s not found in %s" % (cls, HEADERS[cls])
    ref_funcs, ref_data = {}, {}
    aliases = {m.group(1): _norm_type(m.group(2)) for b in ref_bodies for m in re.finditer(r"\busing\s+(\w+)\s*=\s*([^;]+);", b)}
    for b in ref_bodies:
        for k, v in _functions(_without_nested(b) + " " + b).items():
            ref_funcs.setdefault(k, []).extend(v)
        ref_data.update(_data_members(_without_nested(b)))
        for nb in re.finditer(r"\bstruct\s+\w+\s*\{", b):    # members of nested structs (stereoinfo_, scalepyrinfo_, ...)
            inner = _class_body(b[nb.start():], nb.group(0).split()[1].rstrip("{"))
            if inner:
                ref_data.update(_data_members(_without_nested(inner)))
    missing = []
    mock_flat = _without_nested(mock_body)
    for name, sigs in _functions(mock_flat).items():
        if name in (cls, "~" + cls):
            continue
        for ret, ptypes in sigs:
            if (cls, name) in MACRO_MADE:
                if not re.search(MACRO_MADE[(cls, name)], ref):
                    missing.append("%s: the macro line that generates it is gone" % name)
                continue
            cands = ref_funcs.get(name, [])
            if not any(p == ptypes and (r == ret or not ret) for r, p in cands):
                missing.append("%s %s(%s)  -- reference has: %s" % (ret, name, ", ".join(ptypes), cands[:3]))
    mock_data = dict(_data_members(mock_flat))
    for nb in re.finditer(r"\bstruct\s+(\w+)\s*\{", mock_body):
        inner = _class_body(mock_body[nb.start():], nb.group(1))
        if inner:
            mock_data.update(_data_members(_without_nested(inner)))
    for name, ty in mock_data.items():
        if (cls, name) in KNOWN:
            continue
        if name not in ref_data:
            missing.append("data member %s %s: not declared in the reference" % (ty, name))
        elif ref_data[name] != ty and aliases.get(ref_data[name], ref_data[name]) != ty:  # (the same alias name on both sides is a match)
            missing.append("data member %s: mock %s, reference %s" % (name, ty, ref_data[name]))
    assert not missing, "\n".join(missing)
