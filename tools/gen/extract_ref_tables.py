#!/usr/bin/env python3
"""Pins the INTEGER half of the oracle to the reference itself.

Runs in the build container only (it reads /root/reference as TEXT -- the reference cannot be compiled here: its
SrookCppLibraries / Boost dependencies are absent).  It copies no source: it parses the reference's constant tables and
interprets the literal emit statements of its header writer, and writes DATA ONLY to tests/golden/ref_tables.json:

  * `ZZ`, Tables K.1 / K.2 and the MARKER values                     src/jpezy.hpp:36-45, 131-152, 47-127
  * Tables K.3-K.6 as the reference holds them -- (size, code) arrays in the `run * 10 + s + (run == 15)` layout, the EOB / ZRL
    indices -- and the four DHT segments                               src/encoder/huffman_table.hpp:26-282
  * the byte sequence `jpezy_writer::write_header` / `write_eoi` emit  src/encoder/jpezy_writer.hpp:20-105
    for the two property sets the CLI builds (src/encoder/encode_io.hpp:144-160 colour, :177-188 gray) and a few sizes

tests/test_ref_tables.py then holds the oracle, the host codec and (on the GPU) the device entropy coder to these numbers.

    python tools/gen/extract_ref_tables.py [--ref /root/reference] [--out tests/golden/ref_tables.json]
"""
import argparse
import json
import re
from pathlib import Path


def strip_comments(t):
    t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
    return re.sub(r"//[^\n]*", "", t)


def int_array(text, name):
    """the integers of `NAME{ ... };` / `NAME{ { ... } };` (decimal, hex, or `0x..._byte` literals)"""
    m = re.search(r"\b" + re.escape(name) + r"\s*\{(.*?)\}\s*;", text, flags=re.S)
    if not m:
        raise SystemExit(f"array {name} not found")
    body = m.group(1).replace("{", " ").replace("}", " ")
    return [int(tok.replace("_byte", ""), 0) for tok in re.findall(r"0x[0-9a-fA-F]+(?:_byte)?|\d+", body)]


def int_const(text, name):
    m = re.search(r"\b" + re.escape(name) + r"\s*=\s*(0x[0-9a-fA-F]+|\d+)\s*;", text)
    if not m:
        raise SystemExit(f"constant {name} not found")
    return int(m.group(1), 0)


def enum_values(text, name):
    m = re.search(r"SROOK_STRONG_ENUM_BEGIN\(" + name + r"\)\s*\{(.*?)\};", text, flags=re.S)
    out, nxt = {}, 0
    for item in m.group(1).split(","):
        item = item.strip()
        if not item:
            continue
        if "=" in item:
            k, v = item.split("=")
            nxt = int(v.strip(), 0)
            out[k.strip()] = nxt
        else:
            out[item] = nxt
        nxt += 1
    return out


class HeaderInterpreter:
    """Executes the emit statements of jpezy_writer::write_header / write_eoi: `(ofps | ...::KIND) << a << b;` with KIND one of
    Byte (one byte per value), Word (two bytes, big endian), Byte_n(n) (exactly n bytes of a string, zero padded) and Bytes (an
    array as it stands), inside the `if (!comment.empty())` and the three `for` loops the function has."""

    def __init__(self, tables, markers, props):
        self.env = dict(tables)
        self.markers = markers
        self.props = props
        self.out = bytearray()

    def value(self, expr, local):
        e = expr.strip()
        e = re.sub(r"static_cast<[^>]*>", "", e)
        e = re.sub(r"pr\.get<property::At::(\w+)>\(\)", lambda m: f"P[{m.group(1)!r}]", e)
        e = re.sub(r"MARKER::(\w+)", lambda m: str(self.markers[m.group(1)]), e)
        e = e.replace(".size()", ".__len__()").replace(".data()", "").replace(".empty()", ".__len__() == 0")
        e = re.sub(r"\(([^()?]*)\?([^():]*):([^()]*)\)", r"((\2) if (\1) else (\3))", e)      # the one C conditional: (i == 0 ? 0 : 0x11)
        scope = {"P": self.props, **self.env, **local}
        return eval(e, {"__builtins__": {}}, scope)          # integers, table lookups, property fields: nothing else occurs

    def emit(self, kind, values, local):
        for v in values:
            x = self.value(v, local)
            if kind == "Byte":
                self.out.append(int(x) & 0xFF)
            elif kind == "Word":
                self.out += int(x).to_bytes(2, "big")
            elif kind.startswith("Byte_n"):
                n = int(self.value(kind[kind.index("(") + 1:kind.rindex(")")], local))
                b = x.encode() if isinstance(x, str) else bytes(x)
                self.out += (b + bytes(n))[:n]
            elif kind == "Bytes":
                self.out += bytes(x)
            else:
                raise SystemExit(f"unknown stream manipulator {kind}")

    def run(self, body, local=None):
        local = dict(local or {})
        pos = 0
        stmt = re.compile(r"\(ofps\s*\|\s*srook::io::jpeg::bofstream::")
        ctrl = re.compile(r"\b(if|for)\s*\(")
        while pos < len(body):
            ms, mc = stmt.search(body, pos), ctrl.search(body, pos)
            if mc and (not ms or mc.start() < ms.start()):
                # control statement: find its parenthesised head and braced block
                i, depth = mc.end(), 1
                while depth:
                    depth += {"(": 1, ")": -1}.get(body[i], 0)
                    i += 1
                head = body[mc.end():i - 1]
                if body[i:].lstrip()[0] != "{":                    # a single statement without braces
                    k = body.index(";", i) + 1
                    block = body[i:k]
                else:
                    j = body.index("{", i)
                    k, depth = j + 1, 1
                    while depth:
                        depth += {"{": 1, "}": -1}.get(body[k], 0)
                        k += 1
                    block = body[j + 1:k - 1]
                if mc.group(1) == "if":
                    cond = head.strip()
                    if "ofps" in cond and "pr." not in cond:       # `if (!ofps) throw ...`: the stream is good
                        pass
                    else:
                        neg = cond.startswith("!")
                        if bool(self.value(cond.lstrip("!"), local)) != neg:
                            self.run(block, local)
                else:
                    init, cond, _ = [p.strip() for p in head.split(";")]
                    var = re.search(r"(\w+)\s*=", init).group(1)
                    i0 = int(self.value(init.split("=")[1], local))
                    hi = self.value(re.split(r"<", cond, 1)[1], local)
                    for it in range(i0, int(hi)):
                        self.run(block, {**local, var: it})
                pos = k
                continue
            if not ms:
                break
            i, depth = ms.end(), 1                                 # the manipulator, up to the parenthesis that closes `(ofps | ...`
            while depth:
                depth += {"(": 1, ")": -1}.get(body[i], 0)
                i += 1
            kind = body[ms.end():i - 1].strip()
            end = body.index(";", i)
            values = body[i:end].strip()
            assert values.startswith("<<"), values
            self.emit(kind, values[2:].split("<<"), local)
            pos = end + 1
        return self


def function_body(text, name):
    m = re.search(r"\b" + name + r"\s*\(\s*\)[^{;]*\{", text, flags=re.S)
    i, depth = m.end(), 1
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i - 1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=str(Path(__file__).resolve().parents[2] / "tests" / "golden" / "ref_tables.json"))
    a = ap.parse_args()
    ref = Path(a.ref)
    jp = strip_comments((ref / "src/jpezy.hpp").read_text())
    ht = strip_comments((ref / "src/encoder/huffman_table.hpp").read_text())
    wr = strip_comments((ref / "src/encoder/jpezy_writer.hpp").read_text())

    tables = {"ZZ": int_array(jp, "ZZ"), "YQuantumTb": int_array(jp, "YQuantumTb"), "CQuantumTb": int_array(jp, "CQuantumTb")}
    markers = enum_values(jp, "MARKER")
    units = enum_values(jp, "Units")
    huff = {k: int_array(ht, k) for k in ("YDcSizeT", "YDcCodeT", "CDcSizeT", "CDcCodeT", "YAcSizeT", "YAcCodeT", "CAcSizeT", "CAcCodeT")}
    huff.update({k: int_const(ht, k) for k in ("YEOBidx", "YZRLidx", "CEOBidx", "CZRLidx", "DcDhtLength", "AcDhtLength")})
    dht = {k: int_array(ht, k) for k in ("YDcDht", "CDcDht", "YAcDht", "CAcDht")}
    assert len(tables["ZZ"]) == 64 and sorted(tables["ZZ"]) == list(range(64))
    assert [len(huff[k]) for k in ("YDcSizeT", "YAcSizeT", "CAcCodeT")] == [12, 162, 162]
    assert [len(dht[k]) for k in ("YDcDht", "CDcDht", "YAcDht", "CAcDht")] == [33, 33, 183, 183]

    # the two property sets the reference's CLI builds: encode_io.hpp:144-160 (colour; named arguments) and :177-188 (gray)
    eio = strip_comments((ref / "src/encoder/encode_io.hpp").read_text())
    comments = re.findall(r'"(Encoded by [^"]*)"', eio)
    assert len(comments) == 2, comments
    dens = [int(x) for x in re.search(r"dots_inch,\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)", eio).groups()]     # HDensity, VDensity, HThumbnail, VThumbnail
    body_h, body_e = function_body(wr, "write_header"), function_body(wr, "write_eoi")
    headers = []
    for W, H, comment in [(512, 512, comments[0]), (4096, 4096, comments[0]), (7680, 4320, comments[1]), (17, 33, comments[1]), (1920, 1080, ""),
                          (65535, 1, comments[0])]:
        props = {"HSize": W, "VSize": H, "Dimension": 3, "SamplePrecision": 8, "Comment": comment, "Units": units["dots_inch"],
                 "HDensity": dens[0], "VDensity": dens[1], "HThumbnail": dens[2], "VThumbnail": dens[3]}
        env = {**tables, **dht}
        hb = HeaderInterpreter(env, markers, props).run(body_h).out
        eb = HeaderInterpreter(env, markers, props).run(body_e).out
        headers.append({"W": W, "H": H, "comment": comment, "header": list(hb), "eoi": list(eb)})
    assert len(headers[0]["header"]) == 644 and headers[0]["eoi"] == [0xFF, 0xD9], (len(headers[0]["header"]), headers[0]["eoi"])

    out = {
        "_what": "integer tables and header bytes of falgon/jpezy, extracted by tools/gen/extract_ref_tables.py from the reference's source TEXT (data only)",
        "_sources": {"zz_qt_markers": "src/jpezy.hpp:36-45,131-152,47-127", "huffman": "src/encoder/huffman_table.hpp:26-282",
                     "header": "src/encoder/jpezy_writer.hpp:20-105", "properties": "src/encoder/encode_io.hpp:144-160,177-188"},
        "zz": tables["ZZ"], "qt_luma": tables["YQuantumTb"], "qt_chroma": tables["CQuantumTb"], "markers": markers,
        "huffman": huff, "dht": dht, "cli_comments": {"color": comments[0], "gray": comments[1]},
        "cli_property": {"units": units["dots_inch"], "hdensity": dens[0], "vdensity": dens[1], "hthumbnail": dens[2], "vthumbnail": dens[3]},
        "headers": headers,
    }
    Path(a.out).write_text(json.dumps(out, separators=(",", ":")) + "\n")
    print("wrote", a.out, Path(a.out).stat().st_size, "bytes;", len(headers), "headers of", sorted({len(h["header"]) for h in headers}), "bytes")


if __name__ == "__main__":
    main()
