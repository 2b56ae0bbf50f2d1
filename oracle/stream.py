"""CPU restatement of the "LMG3" stream container (include/limg_hip.h) -- TEST INFRASTRUCTURE ONLY, like everything under oracle/.

Upstream has no serialised format, so there is no reference bitstream to compare with; what pins this
container is the round trip: `decode(pack(oracle encode))` must equal the pDecoded plane of the reference / oracle bit for bit,
and the GPU packer must produce these exact bytes from the same image.  numpy + per-block calls into the C oracle's a16
(`limg_oracle_block_decode`, src/limg_decode.h:36-236)."""
import numpy as np

MAGIC = 0x33474D4C
VERSION = 1
HEADER = np.dtype([("magic", "<u4"), ("version", "<u4"), ("sizeX", "<u4"), ("sizeY", "<u4"), ("channels", "<u4"), ("errorFactor", "<u4"),
                   ("blocksX", "<u4"), ("blocksY", "<u4"), ("payloadWords", "<u8"), ("totalBytes", "<u8"), ("flags", "<u4"), ("reserved", "<u4", 3)])
BLOCK = np.dtype([("dirA_min", "<i2", 4), ("dirA_max", "<i2", 4), ("dirB_offset", "<i2", 4), ("dirB_mag", "<i2", 4), ("dirC_offset", "<i2", 4),
                  ("dirC_mag", "<i2", 4), ("shift", "<u4"), ("payloadWord", "<u4")])
assert HEADER.itemsize == 64 and BLOCK.itemsize == 56
VECS = ("dirA_min", "dirA_max", "dirB_offset", "dirB_mag", "dirC_offset", "dirC_mag")


def _field_bits(shift3, rec, channels):
    """bits per pixel of the three fields and the raw-escape mask (see include/limg_hip.h)."""
    bits, raw = [], 0
    pairs = (("dirA_min", "dirA_max"), ("dirB_offset", "dirB_mag"), ("dirC_offset", "dirC_mag"))
    for k in range(3):
        s = int(shift3[k])
        b = 0 if s >= 8 else 8 - s
        if s >= 8 and channels == 4 and int(rec[pairs[k][0]][3]) != int(rec[pairs[k][1]][3]):
            b = 8
            raw |= 1 << k
        bits.append(b)
    return bits, raw


def _pack_field(values8x8, b):
    """8x8 uint values (< 2**b) -> 8*b bytes, pixel (r, x) at bit (8 r + x) * b, little endian."""
    acc = 0
    flat = values8x8.reshape(-1)
    for i in range(64):
        acc |= int(flat[i]) << (i * b)
    return acc.to_bytes(8 * b, "little")


def _unpack_field(buf, b):
    acc = int.from_bytes(bytes(buf), "little")
    mask = (1 << b) - 1
    return np.array([(acc >> (i * b)) & mask for i in range(64)], dtype=np.uint8).reshape(8, 8)


def pack(enc, size_x, size_y, channels, error_factor=100, flags=1):
    """enc: dict from Oracle.encode3d(..., extras=True) (planes + records + shifts + preA/B/C) -> stream bytes (numpy uint8)."""
    bx, by = (size_x + 7) // 8, (size_y + 7) // 8
    n = bx * by
    table = np.zeros(n, dtype=BLOCK)
    payload = bytearray()
    planes = (enc["pFactorsA"], enc["pFactorsB"], enc["pFactorsC"])
    pres = (enc["preA"], enc["preB"], enc["preC"])
    for j in range(by):
        for i in range(bx):
            g = j * bx + i
            rec = enc["records"][j, i]
            sh = enc["shifts"][j, i]
            bits, raw = _field_bits(sh, rec, channels)
            for v in VECS:
                table[g][v] = rec[v]
            table[g]["shift"] = int(sh[0]) | (int(sh[1]) << 8) | (int(sh[2]) << 16) | (raw << 24)
            table[g]["payloadWord"] = len(payload) // 8
            y0, x0 = j * 8, i * 8
            for k in range(3):
                b = bits[k]
                if b == 0:
                    continue
                grid = np.zeros((8, 8), dtype=np.uint32)
                if (raw >> k) & 1:
                    src = pres[k][y0:y0 + 8, x0:x0 + 8]          # the un-dithered factor byte (what the reference's decoder multiplies at shift 8)
                    grid[:src.shape[0], :src.shape[1]] = src
                else:
                    src = planes[k][y0:y0 + 8, x0:x0 + 8]        # plane byte = value << shift
                    grid[:src.shape[0], :src.shape[1]] = src >> (8 - b)
                payload += _pack_field(grid, b)
    hdr = np.zeros(1, dtype=HEADER)
    hdr["magic"], hdr["version"] = MAGIC, VERSION
    hdr["sizeX"], hdr["sizeY"], hdr["channels"], hdr["errorFactor"] = size_x, size_y, channels, error_factor
    hdr["blocksX"], hdr["blocksY"] = bx, by
    hdr["payloadWords"] = len(payload) // 8
    hdr["totalBytes"] = 64 + 56 * n + len(payload)
    hdr["flags"] = flags
    return np.frombuffer(hdr.tobytes() + table.tobytes() + bytes(payload), dtype=np.uint8).copy()


def parse(stream):
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    hdr = stream[:64].view(HEADER)[0]
    assert hdr["magic"] == MAGIC and hdr["version"] == VERSION
    n = int(hdr["blocksX"]) * int(hdr["blocksY"])
    table = stream[64:64 + 56 * n].view(BLOCK)
    payload = stream[64 + 56 * n:int(hdr["totalBytes"])]
    return hdr, table, payload


def decode(stream, oracle):
    """stream -> decoded (h, w) uint32 image, a16 per block through the C oracle."""
    from .bind import REC_DTYPE
    hdr, table, payload = parse(stream)
    w, h, ch = int(hdr["sizeX"]), int(hdr["sizeY"]), int(hdr["channels"])
    bx, by = int(hdr["blocksX"]), int(hdr["blocksY"])
    out = np.zeros((h, w), dtype=np.uint32)
    for j in range(by):
        for i in range(bx):
            e = table[j * bx + i]
            sw = int(e["shift"])
            shift = [sw & 0xFF, (sw >> 8) & 0xFF, (sw >> 16) & 0xFF]
            raw = sw >> 24
            rec = np.zeros(1, dtype=REC_DTYPE)
            for v in VECS:
                rec[v] = e[v]
            rx, ry = min(8, w - i * 8), min(8, h - j * 8)
            o = int(e["payloadWord"]) * 8
            facs = []
            for k in range(3):
                b = 8 if (raw >> k) & 1 else (0 if shift[k] >= 8 else 8 - shift[k])
                grid = _unpack_field(payload[o:o + 8 * b], b) if b else np.zeros((8, 8), dtype=np.uint8)
                o += 8 * b
                facs.append(np.ascontiguousarray(grid[:ry, :rx]).reshape(-1))   # the reference indexes factors by i = yy * rx + xx
            dec = oracle.block_decode(rx, ry, ch, rec, facs[0], facs[1], facs[2], shift)
            out[j * 8:j * 8 + ry, i * 8:i * 8 + rx] = np.asarray(dec, dtype=np.uint32).reshape(ry, rx)
    return out
