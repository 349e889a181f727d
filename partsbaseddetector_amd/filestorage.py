"""FileStorageModel: the reference's on-disk model format (src/FileStorageModel.cpp:42-159), i.e. an
OpenCV ``cv::FileStorage`` document in YAML or XML with the keys

    name, interval, thresh, sbin, norient, flen,
    filtersw (sequence of opencv-matrix, each k x (k*flen), double),
    biasw (float sequence), anchors (Point sequence), defs (sequence of 4-float sequences),
    indexers/component-<c>/part-<p>/{parentid, filterid, biasid, defid}

The arithmetic-free part of the path, but it is what feeds it: a real ``Person_26parts.xml`` drops in
through ``deserialize``.  OpenCV is not available here and the reference ships no model file, so the
reader is written against the format OpenCV's writer produces as documented (YAML 1.0 block/flow
subset with ``!!opencv-matrix``; XML ``<opencv_storage>`` with ``type_id="opencv-matrix"``) and is
exercised on files produced by ``serialize`` below plus hand-written XML ("parity unpinned").

Deliberate difference from the reference: ``defid`` is read as what the writer
wrote -- scalar, sequence or empty -- not through the fork's ``isInt()`` shortcut that collapses every
multi-mixture ``defid`` to ``[0]`` (src/FileStorageModel.cpp:148-152).
"""
from __future__ import annotations

import re
import xml.etree.ElementTree as ET
from typing import Any, List

import numpy as np

from .model import Model


# ------------------------------------------------------------------------------------------ writer
def _fmt(v) -> str:
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    r = repr(float(v))
    if "e" not in r and "." not in r and "inf" not in r and "nan" not in r:
        r += "."
    return r.replace("inf", ".Inf").replace("nan", ".Nan")


def _seq(vals) -> str:
    return "[ " + ", ".join(_fmt(v) for v in vals) + " ]" if len(vals) else "[]"


def serialize(model: Model, filename: str) -> bool:
    """FileStorageModel::serialize (src/FileStorageModel.cpp:42-94), YAML flavour."""
    out = ["%YAML:1.0", f'name: "{model.name}"', f"interval: {model.interval}", f"thresh: {_fmt(float(model.thresh))}",
           f"sbin: {model.sbin}", f"norient: {model.norient}", f"flen: {model.flen}", "filtersw:"]
    for f in model.filtersw:
        f = np.asarray(f, np.float64)
        out += ["   - !!opencv-matrix", f"     rows: {f.shape[0]}", f"     cols: {f.shape[1]}", "     dt: d",
                "     data: " + _seq(f.ravel())]
    out.append("biasw: " + _seq([float(np.float32(b)) for b in model.biasw]))
    out.append("anchors: " + _seq([int(v) for a in model.anchors for v in a]))
    out.append("defs:")
    for d in model.defw:
        out.append("   - " + _seq([float(np.float32(v)) for v in d]))
    out.append("indexers:")
    for c in range(model.ncomponents()):
        out.append(f"   component-{c}:")
        for p in range(model.nparts(c)):
            out.append(f"      part-{p}:")
            out.append(f"         parentid: {model.parentid[c][p]}")
            out.append("         filterid: " + _seq(model.filterid[c][p]))
            out.append("         biasid: " + _seq(model.biasid[c][p]))
            out.append("         defid: " + _seq(model.defid[c][p]))
    with open(filename, "w") as fh:
        fh.write("\n".join(out) + "\n")
    return True


def serialize_xml(model: Model, filename: str) -> bool:
    """The same document in the layout OpenCV's XML writer produces (`fs.open("model.xml", WRITE)`): an
    <opencv_storage> root, sequences of <_> items, `type_id="opencv-matrix"` nodes, numbers wrapped over lines.
    The reference's configs name XML models (conf/config_person.by_parts:30, conf/config_face.by_parts:31)."""
    def wrap(vals, indent, per_line=4):
        toks = [_fmt(v) for v in vals]
        pad = " " * indent
        return "\n".join(pad + " ".join(toks[i:i + per_line]) for i in range(0, len(toks), per_line))
    out = ['<?xml version="1.0"?>', "<opencv_storage>", f'<name>"{model.name}"</name>' if " " in model.name else f"<name>{model.name}</name>",
           f"<interval>{model.interval}</interval>", f"<thresh>{_fmt(float(model.thresh))}</thresh>", f"<sbin>{model.sbin}</sbin>",
           f"<norient>{model.norient}</norient>", f"<flen>{model.flen}</flen>", "<filtersw>"]
    for f in model.filtersw:
        f = np.asarray(f, np.float64)
        out += ['  <_ type_id="opencv-matrix">', f"    <rows>{f.shape[0]}</rows>", f"    <cols>{f.shape[1]}</cols>", "    <dt>d</dt>",
                "    <data>", wrap(f.ravel(), 6) + "</data></_>"]
    out[-1] += "</filtersw>"
    out += ["<biasw>", wrap([float(np.float32(b)) for b in model.biasw], 2, 6) + "</biasw>"]
    out += ["<anchors>", wrap([int(v) for a in model.anchors for v in a], 2, 10) + "</anchors>"]
    out.append("<defs>")
    for d in model.defw:
        out += ["  <_>", wrap([float(np.float32(v)) for v in d], 4) + "</_>"]
    out[-1] += "</defs>"
    out.append("<indexers>")
    for c in range(model.ncomponents()):
        out.append(f"  <component-{c}>")
        for p in range(model.nparts(c)):
            out.append(f"    <part-{p}>")
            out.append(f"      <parentid>{model.parentid[c][p]}</parentid>")
            for key, vals in (("filterid", model.filterid[c][p]), ("biasid", model.biasid[c][p]), ("defid", model.defid[c][p])):
                out += [f"      <{key}>", wrap(vals, 8, 10) + f"</{key}>"] if len(vals) else [f"      <{key}></{key}>"]
            out[-1] += f"</part-{p}>"
        out[-1] += f"</component-{c}>"
    out[-1] += "</indexers>"
    out.append("</opencv_storage>")
    with open(filename, "w") as fh:
        fh.write("\n".join(out) + "\n")
    return True


# ------------------------------------------------------------------------------------------ YAML reader
def _scalar(tok: str):
    tok = tok.strip()
    if tok.startswith('"') and tok.endswith('"'):
        return tok[1:-1]
    low = tok.lower()
    if low in (".inf", "+.inf"):
        return float("inf")
    if low == "-.inf":
        return float("-inf")
    if low == ".nan":
        return float("nan")
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok


def _flow(text: str):
    """parse a (possibly nested) flow sequence '[ a, [b, c], d ]'"""
    pos = 0

    def parse():
        nonlocal pos
        assert text[pos] == "["
        pos += 1
        items, tok = [], ""
        while True:
            ch = text[pos]
            if ch == "[":
                items.append(parse())
                tok = ""
            elif ch in ",]":
                if tok.strip():
                    items.append(_scalar(tok))
                tok = ""
                if ch == "]":
                    pos += 1
                    return items
                pos += 1
            else:
                tok += ch
                pos += 1

    return parse()


def _yaml_lines(text: str):
    lines = []
    for raw in text.splitlines():
        if raw.startswith("%") or raw.strip() in ("", "---", "..."):
            continue
        lines.append(raw.rstrip())
    # join flow sequences that OpenCV wraps over several lines
    joined, buf, depth = [], "", 0
    for ln in lines:
        if depth == 0:
            buf = ln
        else:
            buf += " " + ln.strip()
        depth += ln.count("[") - ln.count("]")
        if depth == 0:
            joined.append(buf)
    return joined


def _parse_block(lines, i, indent):
    """returns (value, next index): a mapping or sequence whose entries are indented by `indent`"""
    is_seq = lines[i].lstrip().startswith("- ")
    result: Any = [] if is_seq else {}
    while i < len(lines):
        ln = lines[i]
        cur = len(ln) - len(ln.lstrip())
        if cur < indent:
            break
        body = ln.strip()
        if is_seq:
            assert body.startswith("-"), ln
            item = body[1:].strip()
            if item.startswith("!!opencv-matrix"):
                sub, i = _parse_block(lines, i + 1, cur + 2)
                result.append(_matrix(sub))
                continue
            if item.startswith("["):
                result.append(_flow(item))
            elif item == "":
                sub, i = _parse_block(lines, i + 1, cur + 1)
                result.append(sub)
                continue
            else:
                result.append(_scalar(item))
            i += 1
        else:
            key, _, rest = body.partition(":")
            rest = rest.strip()
            if rest.startswith("!!opencv-matrix"):
                sub, i = _parse_block(lines, i + 1, cur + 1)
                result[key] = _matrix(sub)
                continue
            if rest == "":
                if i + 1 < len(lines) and (len(lines[i + 1]) - len(lines[i + 1].lstrip())) > cur:
                    nxt = len(lines[i + 1]) - len(lines[i + 1].lstrip())
                    sub, i = _parse_block(lines, i + 1, nxt)
                    result[key] = sub
                    continue
                result[key] = []
            elif rest.startswith("["):
                result[key] = _flow(rest)
            else:
                result[key] = _scalar(rest)
            i += 1
    return result, i


def _matrix(d):
    dt = str(d.get("dt", "d"))
    dtype = {"d": np.float64, "f": np.float32, "i": np.int32, "u": np.uint8}[dt[-1]]
    return np.asarray(d["data"], dtype).reshape(int(d["rows"]), int(d["cols"])).astype(np.float64)


def _read_yaml(text: str):
    lines = _yaml_lines(text)
    doc, _ = _parse_block(lines, 0, 0)
    return doc


# ------------------------------------------------------------------------------------------ XML reader
def _xml_value(node):
    if node.get("type_id") == "opencv-matrix":
        rows, cols = int(node.find("rows").text), int(node.find("cols").text)
        data = np.asarray(node.find("data").text.split(), np.float64)
        return data.reshape(rows, cols)
    children = list(node)
    if children:
        if all(ch.tag == "_" for ch in children):
            return [_xml_value(ch) for ch in children]
        return {ch.tag: _xml_value(ch) for ch in children}
    text = (node.text or "").strip()
    toks = text.split()
    if len(toks) > 1:
        return [_scalar(t) for t in toks]
    if text.startswith('"'):
        return text.strip('"')
    return _scalar(text) if text else []


def _read_xml(text: str):
    root = ET.fromstring(text)
    return {ch.tag: _xml_value(ch) for ch in root}


# ------------------------------------------------------------------------------------------ model
def _as_list(v) -> List:
    if isinstance(v, list):
        return v
    return [v]


def deserialize(filename: str) -> Model:
    """FileStorageModel::deserialize (src/FileStorageModel.cpp:96-159)."""
    text = open(filename).read()
    doc = _read_xml(text) if text.lstrip().startswith("<?xml") or "<opencv_storage>" in text else _read_yaml(text)
    m = Model(name=str(doc.get("name", "")), interval=int(doc["interval"]), thresh=float(doc["thresh"]),
              sbin=int(doc["sbin"]), norient=int(doc["norient"]), flen=int(doc["flen"]))
    m.filtersw = [np.asarray(f, np.float64) for f in doc["filtersw"]]
    m.biasw = [float(np.float32(b)) for b in _as_list(doc["biasw"])]
    an = _as_list(doc["anchors"])
    if an and isinstance(an[0], list):
        m.anchors = [(int(a[0]), int(a[1])) for a in an]
    else:
        m.anchors = [(int(an[i]), int(an[i + 1])) for i in range(0, len(an), 2)]
    m.defw = [[float(np.float32(v)) for v in d] for d in doc["defs"]]
    comps = doc["indexers"]
    for c in range(len(comps)):
        parts = comps[f"component-{c}"]
        fid, bid, did, par = [], [], [], []
        for p in range(len(parts)):
            part = parts[f"part-{p}"]
            par.append(int(part["parentid"]))
            fid.append([int(v) for v in _as_list(part["filterid"])])
            bid.append([int(v) for v in _as_list(part["biasid"])])
            did.append([int(v) for v in _as_list(part.get("defid", []))])
        m.filterid.append(fid)
        m.biasid.append(bid)
        m.defid.append(did)
        m.parentid.append(par)
    m.validate()
    return m
