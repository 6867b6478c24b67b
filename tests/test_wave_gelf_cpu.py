"""CPU: the wave-cooperative GELF tokeniser (flowgger_amd/csrc/fg_gelf2.hpp) -- the SAME source the gfx950 kernel is built
from -- run lane for lane over the fiber emulation of a wavefront (tests/native/fg_wave_emu.hpp) against the oracle.

The fast form either handles a line (then its row and its entries must be byte-identical to the oracle's Record) or hands it
back for the exact general form (allowed for anything that is not a flat object in GELF producers' spelling -- but NOT for
the well-formed corpus, and never with a wrong result)."""
import numpy as np
import pytest

from flowgger_amd import synth

GELF = 2


@pytest.fixture(scope="module")
def wave():
    import wave_binding

    return wave_binding.WaveHost()


def check(wave, oracle, lines, must_handle=None, **geom):
    lines = [ln.encode("utf-8", "surrogateescape") if isinstance(ln, str) else ln for ln in lines]
    data, offsets = synth.pack(lines) if lines else (np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    pad = np.concatenate([data, np.zeros(64, np.uint8)])
    tab, handled = wave.gelf(pad, offsets, **geom)
    oblob, ooffs = oracle.decode_batch(GELF, data, offsets)
    blob, offs = tab.serialize(GELF, pad, offsets)
    for i in np.nonzero(handled)[0]:
        got = blob[int(offs[i]):int(offs[i + 1])].tobytes()
        want = oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes()
        assert got == want, f"line {i}: {lines[i][:300]!r}\n  wave   {got[:300]!r}\n  oracle {want[:300]!r}"
    # entry slices of handled lines never overlap and stay inside the allocated range
    used = tab.ent_used
    seen = np.zeros(used + 1, np.int32)
    for i in np.nonzero(handled)[0]:
        f, c = int(tab.a["ent_first"][i]), int(tab.a["ent_count"][i])
        assert f + c <= used
        seen[f:f + c] += 1
    assert seen.max(initial=0) <= 1
    if must_handle is not None:
        for i in must_handle:
            assert handled[i], f"the fast form handed back a line it must handle: {lines[i][:200]!r}"
    return handled


@pytest.mark.parametrize("geom", [dict(lines_per_group=32, tile_cap=12288), dict(lines_per_group=64, tile_cap=22528),
                                  dict(lines_per_group=16, tile_cap=6144), dict(lines_per_group=7, tile_cap=4096),
                                  dict(lines_per_group=48, tile_cap=16384)])
def test_corpus(wave, oracle, geom):
    lines = synth.gelf_lines(6000)
    handled = check(wave, oracle, lines, **geom)
    # BASELINE configs[2] corpus: everything but the deliberately malformed lines (1 %) is fast-form material
    assert handled.mean() > 0.985, handled.mean()


@pytest.mark.parametrize("ent_cap", [96, 700, 2500])
def test_rows_that_are_not_flagged_stay_valid_when_the_entry_table_overflows(wave, oracle, ent_cap):
    """wv::wave_alloc's overflow branch: the lines before the cut keep their slots, so that part of the chunk must be
    committed -- a later tile of the same wave must not be handed the same slots.  (The emulation reserves chunks of 96.)"""
    base = synth.gelf_lines(800, invalid_frac=0)
    rng = np.random.default_rng(ent_cap)
    lines = []
    for i, ln in enumerate(base):  # (the corpus has eight extras on every line: chunks would be used up exactly)
        k = int(rng.integers(0, 6))
        extras = ",".join(f'"_x{j}":{i * 7 + j}' for j in range(k))
        lines.append(ln if k == 0 else (b'{"host":"h%d",%s}' % (i, extras.encode())))
    data, offsets = synth.pack(lines)
    pad = np.concatenate([data, np.zeros(64, np.uint8)])
    tab, handled = wave.gelf(pad, offsets, lines_per_group=8, tile_cap=4096, ent_cap=ent_cap)
    oblob, ooffs = oracle.decode_batch(GELF, data, offsets)
    st = tab.status
    over = st == 0xFE
    assert over.any() and not over.all()
    blob, offs = tab.serialize(GELF, pad, offsets)
    seen = np.zeros(ent_cap + 1, np.int32)
    for i in np.nonzero(handled & ~over)[0]:
        f, c = int(tab.a["ent_first"][i]), int(tab.a["ent_count"][i])
        assert c == 0 or f + c <= ent_cap, f"line {i} owns slots beyond ent_cap"
        seen[f:f + c] += 1
        got = blob[int(offs[i]):int(offs[i + 1])].tobytes()
        want = oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes()
        assert got == want, f"line {i} is not flagged FG_ST_OVERFLOW but differs from the oracle"
    assert seen.max(initial=0) <= 1, "two unflagged lines own the same entry slots"


def test_semantics(wave, oracle):
    many = "{" + ",".join(f'"k{(i * 7919) % 100:03d}":{i}' for i in range(100)) + ',"host":"h"}'
    sixty = "{" + ",".join(f'"k{(i * 31) % 60:02d}":{i}' for i in range(60)) + ',"host":"h"}'
    lines = [
        '{"host":"h","b":1,"a":2,"_c":null,"B":true,"a":-3}', "[1,2]", '{"host":"h"} x', '{"a":1}', '{"host":1}',
        '{"host":"h","level":-1}', '{"host":"h","level":1.0}', '{"host":"h","version":1}', '{"host":"h","short_message":1}',
        '{"host":"h","full_message":null}', '{"host":"h","x":{"y":1}}', '{"host":"h\\u00e9\\n\\ud83d\\ude00","timestamp":1}',
        '{"host":"a\nb","timestamp":1}', '{"host":"a\tb"}', '{"host":1,"_x":[]}', "", " ", "null", "true", '"str"', "12", "-",
        "{}", "{ }", '{"host":"h",}', '{,"host":"h"}', '{"host" "h"}', '{"host":"h"', '{"host":"h"}}', "[", "[1,]", "[,1]",
        '{"host":"h","a":[1,{"b":[true,false,null,"x",-1.5e3]}],"c":1}', '{"host":"h","a":[1 2]}', '{"host":"h","a":{"b":1,}}',
        '{"host":"h","a":tru}', '{"host":"h","a":nul}', '{"host":"h","a":falsE}', '{"host":"h","a":01}', '{"host":"h","a":1.}',
        '{"host":"h","a":-}', '{"host":"h","a":1e}', '{"host":"h","a":1e999}', '{"host":"h","a":-1e999}', '{"host":"h","a":1e-999}',
        '{"host":"h","a":18446744073709551615,"b":18446744073709551616,"c":-9223372036854775808,"d":-9223372036854775809}',
        '{"host":"h","a":0.1,"b":1385053862.3072,"c":123456789012345678901234567890.5e-5,"d":-0,"e":-0.0,"f":1E+2}',
        '{"host":"h","timestamp":-5}', '{"host":"h","timestamp":18446744073709551615}', '{"host":"h","timestamp":true}',
        '{"h\\u006fst":"escaped key","_a\\"b":"q","_a\\u0022b":"dup wins","a\\\\b":1,"a\\/b":2,"a\\tb":3}',
        '{"host":"h","s":"\\ud83d"}', '{"host":"h","s":"\\ud83d\\u0041"}', '{"host":"h","s":"\\udc00"}', '{"host":"h","s":"\\u12g4"}',
        '{"host":"h","s":"\\x"}', '{"host":"h","s":"abc', '{"host":"h","s":"\\', '{"host":"h","s":"\\u00"}', '{"host":"h","s":"\\u0000"}',
        '{"host":"h","version":"1.0"}', '{"host":"h","version":"1.\\u0031"}', '{"host":"h","version":"1.2"}', '{"host":"h","level":7}',
        '{"host":"h","level":8}', '{"host":"h","level":"1"}', '{"_":1,"__":2,"":3,"host":"h"}',
        '{"host":"line1\nline2\nline3","short_message":"with\nnewline","_k\ney":"v"}', '{"host":"a\nb",\n"x":1}', '{"host":"a\\\nb"}',
        '{"host":"a\nb\\\nc","k\\\n":"v\\\n"}', '{"host":"a\nb","s":"\\ud83d\n\\ude00"}', '{"host":"a\rb"}', '{"host":"a\x00b"}',
        '\n{"host":"h"}\n', '{"host":"h","a":"x\x7fy"}', '{"host":"h","a":"caf\u00e9 \u4e2d\u6587 \U0001F600"}',
        many, sixty, '{"host":"h","big":' + "9" * 400 + "}", '{"host":"h","big":0.' + "9" * 400 + "}",
        '{"host":"h","e":1e2147483648}', '{"host":"h","e":0e2147483648}', '{"host":"h","e":1e-2147483649}',
        '{"zz":[],"host":5}', '{"timestamp":"x","level":99,"host":"h"}', '{"version":"9","timestamp":"x","host":"h"}',
        # spacing, ordering and duplicates the fast form must get right itself
        ' {  "host" : "h" ,  "b" :  1 , "a":"x"  }  ', '{"host":"h","dup":1,"dup":2,"dup":"three"}', '{"host":"a","host":"b","host":"c"}',
        '{"abcdefg1":1,"abcdefg2":2,"host":"h"}', '{"abcdefgh":1,"abcdefgh":2,"host":"h"}', '{"abcdefg":1,"abcdefgh":2,"host":"h"}',
        '{"level":3,"level":"bad","host":"h"}', '{"level":"bad","level":3,"host":"h"}', '{"host":"h","version":"1.1","version":7}',
        '{"timestamp":1.5,"timestamp":2,"host":"h"}', '{"host":"h","q":"say \\"hi\\"","p":"back\\\\slash","r":"a\\\\"}',
        '{"host":"h","k":"v\\\\","l":"\\\\\\"x"}', '{"host":"h","a":"}","b":",","c":"{","d":"]:["}', '{"host":"h","a":"x"x}',
        '{"host":"h" "a":1}', '{"host":"h","a":1 2}', '{"host":"h","a": }', '{"host":"h","":""}', '{"host":"h","a":"\\u00e9\\ud83d\\ude00"}',
        '{"host":"h","a":truex}', '{"host":"h","a":1,"b"}', '{"host":"h","a":1,"b":}', '{"host":"h","a":1,:2}', '{"host":"h"},',
        '{"host":"h","a":1.5e3,"b":-0.0001,"c":1e-7,"d":12345678901234567890,"e":0}', '{"host":"h","level":0}', '{"host":"h","level":007}',
        '{"host":"h","x":"' + "y" * 3000 + '"}',
        # strings whose closing quote lies in the member's second 64 bytes; one / two / odd escapes
        '{"host":"h","short_message":"' + "m" * 70 + '"}', '{"host":"h","short_message":"' + "m" * 70 + '\\n tail"}',
        '{"host":"h","short_message":"' + "m" * 70 + '"  }', '{"host":"h","short_message":"' + "m" * 70 + '" x}',
        '{"host":"h","short_message":"' + "m" * 70 + '\\q"}', '{"host":"h","short_message":"\\t' + "m" * 70 + '"}',
        '{"host":"h","short_message":"' + "m" * 110 + '"}', '{"host":"h","short_message":"' + "m" * 130 + '"}',
        '{"host":"h","a":"x\\\\"}', '{"host":"h","a":"\\\\\\n"}', '{"host":"h","a":"\\n\\t"}', '{"host":"h","a":"\\q"}',
        '{"host":"h","a":"\\/"}', '{"host":"h","a":"tab\\there"}', '{"host":"h","a":"\\u00e9"}', '{"host":"h","a":"\\"}', '{"' + "k" * 255 + '":1,"host":"h"}', '{"' + "k" * 256 + '":1,"host":"h"}',
    ]
    handled = check(wave, oracle, lines)
    by_line = dict(zip(lines, handled))  # (keyed by the str spelling above)
    # producers' spelling is fast-form material
    for ln in ('{"host":"h","b":1,"a":2,"_c":null,"B":true,"a":-3}', "{}", ' {  "host" : "h" ,  "b" :  1 , "a":"x"  }  ',
               '{"host":"h","dup":1,"dup":2,"dup":"three"}', '{"level":"bad","level":3,"host":"h"}', sixty,
               '{"host":"h","q":"say \\"hi\\"","p":"back\\\\slash","r":"a\\\\"}', '{"host":"h","a":"}","b":",","c":"{","d":"]:["}',
               '{"timestamp":"x","level":99,"host":"h"}', '{"host":"h","level":8}'):
        assert by_line[ln], ln
    # ... and these are not: they must be handed back, never guessed
    for ln in ('{"host":"h","x":{"y":1}}', '{"abcdefg1":1,"abcdefg2":2,"host":"h"}', '{"host":"a\nb","timestamp":1}', many,
               '{"host":"h","version":"1.\\u0031"}', '{"host":"h","a":01}'):
        assert not by_line[ln], ln


def test_row_widths_ties_and_duplicates(wave, oracle):
    """The rows of the item pass: lines of up to 16 items share a wave four at a time and rank their extras by the keys' first four
    bytes (DPP rotations); wider lines get rows of 32 / 64 lanes, and keys the four bytes cannot order, or a key met twice, take
    the 64-bit form through LDS.  All of these are producers' spelling: the fast form must handle every one itself."""
    def obj(n, pfx="_k", host_at=0, extra=""):
        f = [f'"{pfx}{(i * 7) % n:02d}":{i}' for i in range(n)]
        f.insert(host_at % (n + 1), '"host":"h"')
        return "{" + ",".join(f) + extra + "}"

    lines = []
    for n in (0, 1, 2, 12, 13, 14, 15, 16, 17, 29, 30, 31, 32, 33, 47, 61, 62):
        lines.append(obj(n, host_at=n // 2))
    # four-byte ties (ordered by bytes 4..6), seven-byte ties with different lengths handed back, duplicates of extras and known keys
    lines += ['{"_user_name":"n","_user_id":7,"host":"h","_user":1,"_use":2,"_us":3}',
              '{"host":"h","_user_id":1,"_user_id":2,"_a":3}', '{"host":"a","_z":1,"host":"b","_y":2,"host":"c"}',
              '{"level":3,"_b":1,"level":"bad","host":"h","_a":2}', '{"level":"bad","_b":1,"level":3,"host":"h","_a":2}',
              '{"timestamp":"x","_q":1,"level":99,"host":"h","version":"7"}',
              '{"version":"1.1","host":"h","short_message":"m","full_message":"f","timestamp":1.5,"level":3,"_a":1,"_b":2,"_c":3,'
              '"_d":4,"_e":5,"_f":6,"_g":7,"_h":8,"_i":9}',  # 16 items exactly: six known keys + nine extras
              '{"version":"1.1","host":"h","short_message":"m","full_message":"f","timestamp":1.5,"level":3,"_a":1,"_b":2,"_c":3,'
              '"_d":4,"_e":5,"_f":6,"_g":7,"_h":8,"_i":9,"_j":10}']  # 17: a row of 32
    rng = np.random.default_rng(4)
    mixed = [lines[int(i)] for i in rng.integers(0, len(lines), 400)] + synth.gelf_lines(400, invalid_frac=0)
    order = rng.permutation(len(mixed))
    mixed = [mixed[int(i)] for i in order]
    for geom in (dict(), dict(lines_per_group=8, tile_cap=4096), dict(lines_per_group=5, tile_cap=4096),
                 dict(lines_per_group=64, tile_cap=24576)):
        handled = check(wave, oracle, lines, must_handle=range(len(lines)), **geom)
        assert handled.all()
        handled = check(wave, oracle, mixed, **geom)
        assert handled.all()


def test_odd_quotes_do_not_leak_into_the_next_line(wave, oracle):
    """A line with an unbalanced quote flips the raw string parity of everything behind it in the tile: the in-string state
    must start afresh at every line (the toggle pass)."""
    good = ['{"host":"h%d","a":"x,y","n":%d}' % (i, i) for i in range(40)]
    bad = ['{"host":"h","s":"abc', '"', '{"a":"\\"}', 'x"y"z"', '{"host":"h"}"']
    lines = []
    for i, g in enumerate(good):
        lines.append(g)
        if i % 3 == 0:
            lines.append(bad[(i // 3) % len(bad)])
    handled = check(wave, oracle, lines, must_handle=[i for i, ln in enumerate(lines) if ln in good])
    assert handled.sum() >= len(good)


def test_fuzz_mutations(wave, oracle):
    rng = np.random.default_rng(808)
    base = synth.gelf_lines(4000, invalid_frac=0)
    alphabet = [b'"', b"\\", b",", b":", b"{", b"}", b"[", b"]", b" ", b"\n", b"\t", b"0", b"-", b".", b"e", b"u", b"n",
                b"true", b"null", b"", b"\\u00e9", b"\\n", "é".encode(), b"\x01", b"_"]
    lines = []
    for ln in base:
        b = bytearray(ln)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, len(b)))
            if b[pos] >= 0x80:
                continue
            b[pos:pos + 1] = alphabet[int(rng.integers(0, len(alphabet)))]
        lines.append(bytes(b))
    check(wave, oracle, lines)
    check(wave, oracle, lines, lines_per_group=64, tile_cap=24576)


def test_ragged_groups_and_lines_outside_the_tile(wave, oracle):
    ok = '{"host":"h","a":%d,"s":"%s"}'
    lines = [ok % (i, "z" * ((i * 37) % 900)) for i in range(150)] + [""] * 5 + [ok % (1, "q" * 20000)] + [ok % (2, "")] * 3
    check(wave, oracle, lines, lines_per_group=32, tile_cap=8192)
    check(wave, oracle, lines[:1])
    check(wave, oracle, [])


def test_register_number_parser_matches_serde_json(wave, oracle):
    """parse_num24 (the v_dot4 digit fold on a token held in registers) against the oracle's serde_json 0.8 number scanner:
    identical kind and bits whenever it accepts, and it must accept the everyday shapes (else the kernel silently falls back to
    the byte-wise parser and the fast form is not fast)."""
    import ctypes as C

    rng = np.random.default_rng(24)
    toks = ["0", "-0", "7", "42", "-1", "8080", "65535", "4294967295", "4294967296", "9999999999999999999", "1234567890123456789",
            "-9223372036854775808", "-9223372036854775809", "-9223372036854775807", "18446744073709551615", "0.5", "-0.5", "0.0",
            "1385053862.3072", "1438790025.637824", "123456789.123456789", "0.000001", "99999.99999", "1.0", "-12345.678",
            "3.141592653589793", "0.1234567890123456789", "1234567890.123456789", "12.5", "100000000000000000.5"]
    for _ in range(3000):
        ni, nf = int(rng.integers(1, 20)), int(rng.integers(0, 19))
        s = ("-" if rng.random() < 0.3 else "") + str(int(rng.integers(1, 10))) + "".join(str(int(d)) for d in rng.integers(0, 10, ni - 1))
        if rng.random() < 0.5 and nf:
            s += "." + "".join(str(int(d)) for d in rng.integers(0, 10, nf))
        toks.append(s[:24])
    odd = ["01", "1.", ".5", "-", "1e5", "1E-3", "1.5e3", "--1", "1-", "1..2", "1.2.3", "+1", "0x10", "12a", "1 ", "00", "-01", "0.5.", "-.5"]
    accepted = 0
    for t in toks + odd:
        b = t.encode()
        kind, bits = C.c_uint32(), C.c_uint64()
        ok = wave.lib.fgw_parse_num24(b, len(b), C.byref(kind), C.byref(bits))
        want = oracle.json_number(t)
        if ok:
            accepted += 1
            assert want is not None and (kind.value, bits.value) == want, (t, kind.value, hex(bits.value), want)
        elif t in toks:
            digits = sum(ch.isdigit() for ch in t)
            assert digits > 19, f"everyday shape rejected: {t}"
    assert accepted >= sum(1 for t in toks if sum(ch.isdigit() for ch in t) <= 19)


@pytest.mark.parametrize("framing", ["line", "nul"])
@pytest.mark.parametrize("geom", [dict(lines_per_group=8, tile_cap=4096), dict(lines_per_group=32, tile_cap=12288)])
def test_framed_streams(wave, oracle, framing, geom):
    """Frames of a raw stream (terminators still in the tile, between the lines): a terminator is not a control character of either
    neighbour, a control character INSIDE a line still takes it out of the fast form; CRLF, empty frames, an unterminated tail.
    Every handled row equals the oracle's decode of the bare line, and the well-formed corpus stays fast-form material."""
    delim = b"\n" if framing == "line" else b"\0"
    lines = synth.gelf_lines(3000)
    extra = [b'{"host":"h","a":"tab\there"}', b'{"host":"h",\t"a":1}', b'{"host":"a\rb"}', b'{"host":"h","a":"x\x01y"}', b'{"host":"h"}\r',
             b' {"host":"h" , "b" : 1}', b'{"host":"h","s":"' + b"m" * 90 + b'"}', b"{}", b"[1,2]", b""]
    if framing == "nul":
        extra += [b'{"host":"line1\nline2"}', b'{"host":"h",\n"a":1}', b'\n{"host":"h"}\n']
    bare, frames = [], []
    for i, ln in enumerate(lines):
        if i % 11 == 4:
            ln = extra[(i // 11) % len(extra)]
        term = delim
        body = ln
        if framing == "line" and i % 5 == 1 and b"\n" not in ln:
            term = b"\r\n"           # lines(): "\n", then ONE "\r", are not part of the line
        frames.append(body + term)
        whole = body + term
        cut = whole[:-1]                  # BufRead::lines() / split(0): the terminator, then (lines only) ONE "\r"
        if framing == "line" and cut.endswith(b"\r"):
            cut = cut[:-1]
        bare.append(cut)
    frames.append(b'{"host":"tail"}')  # unterminated last frame
    bare.append(b'{"host":"tail"}')
    raw = b"".join(frames)
    offsets = np.zeros(len(frames) + 1, np.uint64)
    offsets[1:] = np.cumsum([len(f) for f in frames])
    pad = np.concatenate([np.frombuffer(raw, np.uint8), np.zeros(64, np.uint8)])
    tab, handled = wave.gelf(pad, offsets, strip=1 if framing == "line" else 2, **geom)
    bdata, boffs = synth.pack(bare)
    oblob, ooffs = oracle.decode_batch(GELF, bdata, boffs)
    blob, offs = tab.serialize(GELF, pad, offsets)
    for i in np.nonzero(handled)[0]:
        got = blob[int(offs[i]):int(offs[i + 1])].tobytes()
        want = oblob[int(ooffs[i]):int(ooffs[i + 1])].tobytes()
        assert got == want, f"frame {i}: {bare[i][:200]!r}\n  wave   {got[:300]!r}\n  oracle {want[:300]!r}"
    corpus = np.array([i % 11 != 4 for i in range(len(lines))] + [True])
    assert handled[corpus].mean() > 0.985, handled[corpus].mean()
    # a control character inside a line is never fast-form material
    for i, b in enumerate(bare):
        if any(c < 0x20 for c in b):
            assert not handled[i], b
