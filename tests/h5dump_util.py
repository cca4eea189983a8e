"""Independent look at the files this build writes: `h5dump -H -A -p` (HDF5's own tool, /opt/conda/bin) parsed into
plain dicts.  Used to hold input, output and checkpoint files to the rules the reference's reader and MATLAB rely on
(Hdf5/Hdf5File.cpp:59-68, 345-352, 767-815, 898-915, 1023-1034; Hdf5FileHeader.cpp:62-87; RealMatrix.cpp:88-121)
without going through the repo's own Hdf5File class."""
import os
import re
import shutil
import subprocess

H5DUMP = shutil.which("h5dump") or "/opt/conda/bin/h5dump"


def available() -> bool:
    return os.path.exists(H5DUMP)


def _tokens(text):
    # braces, quoted strings, and bare words
    return re.findall(r'"[^"]*"|[{}]|[^\s{}"]+', text)


def _parse_block(tok, i):
    """tok[i] is just after '{'; returns (items, index after the matching '}').  items: list of (head words, block or None)"""
    items, head = [], []
    while i < len(tok):
        t = tok[i]
        if t == "{":
            block, i = _parse_block(tok, i + 1)
            items.append((head, block))
            head = []
        elif t == "}":
            if head:
                items.append((head, None))
            return items, i + 1
        else:
            head.append(t)
            i += 1
    return items, i


def _flat(block):
    out = []
    for head, sub in block or []:
        out += head
        if sub is not None:
            out += ["{"] + _flat(sub) + ["}"]
    return out


def _attr(block):
    words = _flat(block)
    text = " ".join(words)
    info = {"string": "H5T_STRING" in words, "scalar": "SCALAR" in text}
    m = re.search(r"STRSIZE (\d+);", text)
    info["strsize"] = int(m.group(1)) if m else None
    info["nullterm"] = "H5T_STR_NULLTERM;" in words
    info["ascii"] = "H5T_CSET_ASCII;" in words
    m = re.search(r"DATATYPE (H5T_\w+)", text)
    info["datatype"] = m.group(1) if m else None
    m = re.search(r'\(0\): (?:"([^"]*)"|(\S+))', text)
    info["value"] = (m.group(1) if m.group(1) is not None else m.group(2)) if m else None
    return info


def _dataset(block):
    d = {"attrs": {}}
    for head, sub in block:
        if head[:1] == ["ATTRIBUTE"]:
            d["attrs"][head[1].strip('"')] = _attr(sub)
            continue
        words = head + (_flat(sub) if sub is not None else [])
        text = " ".join(words)
        if head and head[0] == "DATATYPE":
            d["datatype"] = head[1] if len(head) > 1 else None
        if "DATASPACE" in head:
            m = re.search(r"\( ([\d, ]+) \) /", text)
            d["dims"] = tuple(int(v) for v in m.group(1).split(",")) if m else ()
        if head[:1] == ["STORAGE_LAYOUT"]:
            m = re.search(r"CHUNKED \( ([\d, ]+) \)", text)
            d["chunk"] = tuple(int(v) for v in m.group(1).split(",")) if m else None
        if head[:1] == ["FILTERS"]:
            m = re.search(r"DEFLATE (?:\{ )?LEVEL (\d+)", text)
            d["deflate"] = int(m.group(1)) if m else 0
    return d


def describe(path):
    """{"attrs": root attributes, "datasets": {"/name" or "/group/name": {...}}, "groups": [...]}"""
    text = subprocess.run([H5DUMP, "-H", "-A", "-p", path], check=True, stdout=subprocess.PIPE, text=True).stdout
    tok = _tokens(text)
    # HDF5 "<file>" { GROUP "/" { ... } }
    start = tok.index("GROUP")
    root, _ = _parse_block(tok, tok.index("{", start) + 1)
    out = {"attrs": {}, "datasets": {}, "groups": []}

    def walk(block, prefix):
        for head, sub in block:
            if head[:1] == ["ATTRIBUTE"] and prefix == "":
                out["attrs"][head[1].strip('"')] = _attr(sub)
            elif head[:1] == ["DATASET"]:
                out["datasets"][prefix + "/" + head[1].strip('"')] = _dataset(sub)
            elif head[:1] == ["GROUP"]:
                name = prefix + "/" + head[1].strip('"')
                out["groups"].append(name)
                walk(sub, name)

    walk(root, "")
    return out


def check_kwave_conventions(desc, file_type):
    """The rules every k-Wave file obeys; returns a list of violations (empty = fine)."""
    bad = []
    for name in ("created_by", "creation_date", "file_description", "file_type", "major_version", "minor_version"):
        a = desc["attrs"].get(name)
        if a is None:
            bad.append(f"root attribute {name} missing")
            continue
        # read with H5LTget_attribute_string into a 256-byte buffer (Hdf5File.cpp:1023-1034): fixed-length, NUL-terminated
        if not (a["string"] and a["scalar"] and a["nullterm"] and a["ascii"] and a["strsize"] is not None and
                a["strsize"] <= 256 and a["strsize"] == len(a["value"]) + 1):
            bad.append(f"root attribute {name}: not a fixed-length NUL-terminated ASCII string ({a})")
    if desc["attrs"].get("file_type", {}).get("value") != file_type:
        bad.append(f"file_type is {desc['attrs'].get('file_type', {}).get('value')!r}, expected {file_type!r}")
    if (desc["attrs"].get("major_version", {}).get("value"), desc["attrs"].get("minor_version", {}).get("value")) != ("1", "1"):
        bad.append("file version is not 1.1")
    for name, d in desc["datasets"].items():
        dt, dom = d["attrs"].get("data_type"), d["attrs"].get("domain_type")
        if dt is None or dom is None:
            bad.append(f"{name}: data_type / domain_type attribute missing")
            continue
        for a in (dt, dom):
            if not (a["string"] and a["nullterm"] and a["strsize"] == len(a["value"]) + 1):
                bad.append(f"{name}: attribute is not a fixed-length NUL-terminated string")
        # element types (Hdf5File.cpp:345-352): floats IEEE F32LE, indices STD_U64LE
        want = {"float": "H5T_IEEE_F32LE", "long": "H5T_STD_U64LE"}.get(dt["value"])
        if want is None or d.get("datatype") != want:
            bad.append(f"{name}: data_type {dt['value']!r} stored as {d.get('datatype')}")
        if dom["value"] not in ("real", "complex"):
            bad.append(f"{name}: domain_type {dom['value']!r}")
        if len(d.get("dims", ())) not in (3, 4):
            bad.append(f"{name}: rank {len(d.get('dims', ()))} (k-Wave datasets are 3-D, cuboid series 4-D)")
        if dom["value"] == "complex" and d["dims"][-1] % 2:
            bad.append(f"{name}: complex data needs an even (doubled) fastest dimension")
    return bad
