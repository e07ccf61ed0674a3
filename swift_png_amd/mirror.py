"""Python mirror of the reference's own API for the hot path, delegating to the C ABI.

Names, argument meaning and error behaviour follow the Swift originals so that the parity tests
read like the reference's tests.  State that the reference keeps as a resumable state machine
(LZ77.InflatorState, PNG.Decoder.row/pass) is kept on the device between pushes: the compressed bytes and the
inflated bytes so far, and the point (a block boundary) from which the next push goes on
(spng_inflate_resume_batch).
"""
from __future__ import annotations

from . import (DONE, FORMAT_GZIP, FORMAT_IOS, FORMAT_ZLIB, NEED_MORE_INPUT, DecodingError, SpngError,
               E_EXTRANEOUS_COMPRESSED_DATA, E_INCOMPLETE_DATASTREAM, E_OUTPUT_CAPACITY, IMAGE_OVERDRAW)

_DELAY_FORMATS = {1: (8, 1), 2: (8, 2), 3: (8, 3), 4: (8, 4), 6: (16, 3), 8: (16, 4)}


class LZ77:
    class Format:
        zlib = FORMAT_ZLIB
        ios = FORMAT_IOS

    class Inflator:
        """LZ77.Inflator (Sources/LZ77/Inflator/LZ77.Inflator.swift:8-62).

        Streaming like the original: the compressed bytes pushed so far and the bytes inflated so far stay on the
        device, and a push goes on where the previous one stopped (spng_inflate_resume_batch: blocks the input now
        holds completely are decoded once, by the parallel pipeline; only the block the input ends in is decoded
        again by the next push).  gzip members (Gzip.Inflator) too: their few header bytes are parsed again by every
        push (LZ77.InflatorBuffers.swift:139-230), their DEFLATE payload is not."""

        def __init__(self, format=FORMAT_ZLIB, session=None):
            from . import load
            self._s = session or load()
            self._format = format
            self._streaming = True
            self._in = bytearray()
            self._d_in, self._n_in = None, 0      # device: all compressed bytes so far
            self._d_out = None                    # device: all inflated bytes so far
            self._state = (0, 0, 0, 0)
            self._out = b""
            self._avail = 0
            self._cursor = 0
            self._terminal = False

        def _append(self, data):
            s, t = self._s, self._s.torch
            need = self._n_in + len(data)
            if self._d_in is None or self._d_in.numel() < need:
                grown = s.empty(max(2 * need, 1 << 16))
                if self._n_in:
                    grown[:self._n_in] = self._d_in[:self._n_in]
                self._d_in = grown
            if data:
                self._d_in[self._n_in:need] = s.to_device(bytes(data))
            self._n_in = need

        def push(self, data) -> object:
            """Returns None once a complete stream has been received, () while it wants more."""
            from . import raise_for
            if self._terminal:
                return None                      # .terminal: remaining input is ignored (:38-40)
            if not self._streaming:
                self._in += bytes(data)
                cap = max(1 << 16, 1100 * len(self._in))
                while True:
                    status, out, _, aux = self._s.inflate(bytes(self._in), self._format, cap)
                    if status != E_OUTPUT_CAPACITY:
                        break
                    cap *= 4                     # the reference's output buffer is unbounded
                raise_for(status, aux)
                self._out, self._avail = out, len(out)
                self._terminal = status == DONE
                return None if self._terminal else ()
            self._append(data)
            if self._d_out is None:
                self._d_out = self._s.empty(max(1 << 16, 8 * self._n_in))
            while True:
                res, state = self._s.inflate_resume(self._d_in, self._n_in, self._d_out, self._format, self._state)
                if res.status != E_OUTPUT_CAPACITY:
                    break
                grown = self._s.empty(4 * self._d_out.numel())          # the reference's output buffer is unbounded
                grown[:self._d_out.numel()] = self._d_out
                self._d_out = grown
            raise_for(res.status, (res.aux[0], res.aux[1]))
            self._state, self._avail = state, res.written
            self._terminal = res.status == DONE
            return None if self._terminal else ()

        def _bytes(self, a, b):
            if not self._streaming:
                return self._out[a:b]
            return bytes(self._d_out[a:b].cpu().numpy()) if b > a else b""

        def pull(self, count=None):
            if count is None:                    # pull() -> everything available (:58-61)
                data, self._cursor = self._bytes(self._cursor, self._avail), self._avail
                return data
            if self._avail - self._cursor < count:
                return None                      # pull(_:) -> nil (:53-56)
            data = self._bytes(self._cursor, self._cursor + count)
            self._cursor += count
            return data


    class Deflator:
        """LZ77.Deflator (Sources/LZ77/Deflator/LZ77.Deflator.swift:8-44): init(format:level:exponent:hint:),
        push(_:last:), pull(), pop().  Every push goes to the device (spng_deflate_resume_batch): the input so far and the
        stream so far live in HBM, the compressor's state -- parse position, queued terms, symbol costs, block limit, bit
        writer -- in a device-side state block, and a push emits what the bytes so far determine (the reference compresses
        whenever more than 4096 bytes are buffered, LZ77.DeflatorBuffers.swift:68-93; its output does not depend on how the
        input was pushed, SURVEY 8a row a13, and neither does this one).  pop() hands out complete chunks of 2 * hint bytes
        (the reference's chunk is 2 * ManagedBuffer.capacity >= 2 * hint bytes, a platform-dependent size) and pull() also
        whatever has been written so far -- their concatenation is the reference's stream bit for bit.  One known difference,
        on a reference bug: a row-by-row pushed stream whose last non-final compress() leaves fewer than three bytes queued
        makes the reference emit a corrupt stored tail (DeflatorBuffers.Stream.swift:45-60); this class emits the correct
        stream instead."""

        def __init__(self, format=FORMAT_ZLIB, level=9, exponent=15, hint=1 << 12, session=None):
            from . import load
            if not 8 <= exponent <= 15:
                raise ValueError("exponent must be 8 ... 15")
            self._s = session or load()
            self._format, self._level, self._exponent = format, level, exponent
            self._chunk = 2 * max(int(hint), 1)
            s = self._s
            self._d_in, self._n = s.empty(1 << 16), 0
            self._d_out = s.empty(s.lib.spng_deflate_bound(1 << 16))
            self._state = s.torch.zeros(int(s.lib.spng_deflate_state_bytes()), dtype=s.torch.uint8, device=s.tdev)
            self._hstate = (0, 0)
            self._avail = 0                      # stream bytes the device has produced so far
            self._cursor = 0
            self._done = False
            self.device_calls = 0

        def push(self, data, last=False):
            if self._done:
                raise RuntimeError("push after the last block")
            s, data = self._s, bytes(data)
            need = self._n + len(data)
            if self._d_in.numel() < need:
                grown = s.empty(2 * need); grown[:self._n] = self._d_in[:self._n]; self._d_in = grown
            if data:
                self._d_in[self._n:need] = s.to_device(data)
            self._n = need
            cap = int(s.lib.spng_deflate_bound(need)) + 64
            if self._d_out.numel() < cap:
                grown = s.empty(2 * cap); grown[:self._avail + 8] = self._d_out[:self._avail + 8]; self._d_out = grown
            res, self._hstate = s.deflate_resume(self._d_in, self._n, self._d_out, self._level, self._state, last, self._format,
                                                 self._exponent, self._hstate)
            self.device_calls += 1
            if res.status not in (DONE, NEED_MORE_INPUT):
                raise SpngError(res.status)
            self._avail = res.written
            self._done = bool(last)

        def _bytes(self, a, b):
            return bytes(self._d_out[a:b].cpu().numpy()) if b > a else b""

        def pop(self):
            """A complete chunk, or None (:40-43)."""
            if self._avail - self._cursor < self._chunk:
                return None
            data = self._bytes(self._cursor, self._cursor + self._chunk)
            self._cursor += self._chunk
            return data

        def pull(self):
            """A complete chunk if there is one, else whatever has been written so far, else None (:31-35)."""
            data = self.pop()
            if data is not None:
                return data
            if self._cursor >= self._avail:
                return None
            data, self._cursor = self._bytes(self._cursor, self._avail), self._avail
            return data


class Gzip:
    """Gzip (Sources/LZ77/Gzip/Gzip.swift:1-47): the LZ77 codec behind a gzip member -- header with FEXTRA / FNAME /
    FCOMMENT skipped, CRC-32 checked, ISIZE read (Gzip.StreamHeader.swift:17-97, LZ77.InflatorBuffers.swift:139-230);
    on the way out the fixed ten-byte header and the CRC-32 / byte-count trailer (LZ77.DeflatorBuffers.swift:96-135)."""

    class Inflator(LZ77.Inflator):
        """Gzip.Inflator (Gzip.Inflator.swift:1-58): init(), push(_:), pull(_:), pull()."""

        def __init__(self, session=None):
            super().__init__(FORMAT_GZIP, session)

    class Deflator(LZ77.Deflator):
        """Gzip.Deflator (Gzip.Deflator.swift:1-40): init(level:exponent:hint:), push(_:last:), pull(), pop()."""

        def __init__(self, level, exponent=15, hint=1 << 12, session=None):
            super().__init__(FORMAT_GZIP, level, exponent, hint, session)

    @staticmethod
    def extract(data, session=None) -> bytes:
        """Gzip.extract(from:) (Gzip.swift:5-11)"""
        inflator = Gzip.Inflator(session)
        inflator.push(data)
        return inflator.pull()

    @staticmethod
    def archive(data, level=7, hint=128 << 10, session=None) -> bytes:
        """Gzip.archive(bytes:level:hint:) (Gzip.swift:32-46)"""
        deflator = Gzip.Deflator(level, hint=hint, session=session)
        deflator.push(data, last=True)
        out = b""
        while (part := deflator.pull()) is not None:
            out += part
        return out


class PNG:
    class Standard:
        common = FORMAT_ZLIB
        ios = FORMAT_IOS

    class Decoder:
        @staticmethod
        def defilter(line: bytes, last: bytes, delay: int, session=None) -> bytes:
            """PNG.Decoder.defilter(_:last:delay:) (PNG.Decoder.swift:152-196).  `line` and `last`
            are pitch+1 bytes (index 0 = filter byte).  delay must be a real PNG pixel stride
            (1, 2, 3, 4, 6 or 8)."""
            from . import load
            s = session or load()
            pitch = len(line) - 1
            depth, channels = _DELAY_FORMATS[delay]
            assert pitch % delay == 0
            rows = bytes([0]) + bytes(last[1:]) + bytes(line)
            _, storage = s.unfilter(rows, pitch // delay, 2, depth, channels, False)
            return bytes(line[:1]) + storage[pitch:]

    class Encoder:
        @staticmethod
        def filter(line: bytes, last: bytes, delay: int, session=None) -> bytes:
            """PNG.Encoder.filter(_:last:delay:) (PNG.Encoder.swift:132-204): line[0] == 0."""
            from . import load
            s = session or load()
            pitch = len(line) - 1
            depth, channels = _DELAY_FORMATS[delay]
            assert pitch % delay == 0
            rows = s.filter(bytes(last[1:]) + bytes(line[1:]), pitch // delay, 2, depth, channels, False)
            return rows[pitch + 1:]

    class ImageEncoder:
        """PNG.Encoder driven over a whole image (PNG.Encoder.pull, Sources/PNG/Encoding/PNG.Encoder.swift:33-129,
        as PNG.Image.compress(stream:level:hint:) loops over it, PNG.Image.swift:658-665): every pull returns the
        payload of the next IDAT chunk, None when the stream is exhausted.  Filter selection and DEFLATE run on the
        device in one spng_encode_batch call at the first pull."""

        def __init__(self, storage: bytes, size, depth, channels, interlaced=False, standard=FORMAT_ZLIB, level=9,
                     hint=1 << 15, session=None):
            from . import load
            self._s = session or load()
            self._args = (bytes(storage), size[0], size[1], depth, channels, bool(interlaced))
            self._standard, self._level = standard, level
            self._chunk = 2 * max(int(hint), 1)
            self._out, self._cursor = None, 0

        def pull(self):
            if self._out is None:
                rows = self._s.filter(*self._args)
                self._out = self._s.deflate(rows, self._level, self._standard)
            if self._cursor >= len(self._out):
                return None
            data = self._out[self._cursor:self._cursor + self._chunk]
            self._cursor += len(data)
            return data

    class Context:
        """PNG.Context (Sources/PNG/Decoding/PNG.Context.swift:56-147), image side reduced to the
        fields that cross the boundary: size, pixel depth/channels, interlacing, standard."""

        def __init__(self, size, depth, channels, interlaced=False, standard=FORMAT_ZLIB, session=None):
            from . import load, storage_size, inflated_size
            self._s = session or load()
            self.size, self.depth, self.channels = tuple(size), depth, channels
            self.interlaced, self.standard = bool(interlaced), standard
            self._storage_bytes = storage_size(size[0], size[1], depth, channels)
            self._storage = bytes(self._storage_bytes)
            self._stale = False                  # the device raster is ahead of the host copy
            self._continue = True                # PNG.Decoder.continue (:22-23)
            # device state kept between pushes: the IDAT bytes so far, the inflated scanlines so far (they are the next
            # push's LZ77 window: never defiltered in place), the defiltered scanlines of interlaced / sub-byte images, the
            # raster; and PNG.Decoder.row / pass as the number of inflated bytes already defiltered
            s = self._s
            self._U = inflated_size(size[0], size[1], depth, channels, self.interlaced)
            self._d_idat, self._n_idat = None, 0
            self._d_rows = s.empty(self._U + 64)     # (64 bytes of slack: a little too much data shows as `written > U`)
            direct = not self.interlaced and depth * channels >= 8
            self._d_work = None if direct else s.empty(self._U + 64)
            self._d_storage = s.to_device(self._storage) if self._storage_bytes else s.empty(1)
            self._state = (0, 0, 0, 0)
            self._defiltered = 0
            self.defiltered_total = 0            # scanline bytes handed to the defilter over all pushes (each row once: == U at the end)

        @property
        def storage(self) -> bytes:
            """PNG.Image.storage so far (fetched from the device when it is asked for)"""
            if self._stale:
                self._storage = bytes(self._d_storage[:self._storage_bytes].cpu().numpy())
                self._stale = False
            return self._storage

        def push(self, data: bytes, overdraw: bool = False):
            """push(data:overdraw:) (:88-102): one call per IDAT chunk.  overdraw: the assigned pixels of an unfinished interlaced
            image are replicated over the cells the later passes refine (PNG.Image.overdraw, PNG.Image.swift:134-183), on the
            device (SPNG_IMAGE_OVERDRAW).  The inflate goes on where the previous chunk stopped
            (spng_inflate_resume_batch); the scanlines that became complete with this chunk -- and only those -- are defiltered
            and assigned (spng_unfilter_resume_batch, PNG.Decoder.swift:88-94, 121-135)."""
            from . import raise_for, E_EXTRANEOUS_IMAGE_DATA
            if not self._continue:
                raise DecodingError(E_EXTRANEOUS_COMPRESSED_DATA)     # PNG.Decoder.swift:51-55
            s = self._s
            need = self._n_idat + len(data)
            if self._d_idat is None or self._d_idat.numel() < need:
                grown = s.empty(max(2 * need, 1 << 16))
                if self._n_idat:
                    grown[:self._n_idat] = self._d_idat[:self._n_idat]
                self._d_idat = grown
            if data:
                self._d_idat[self._n_idat:need] = s.to_device(bytes(data))
            self._n_idat = need
            res, state = s.inflate_resume(self._d_idat, self._n_idat, self._d_rows, self.standard, self._state)
            status = res.status
            # PNG.Decoder.swift:142-147: anything left in the inflator after the last row
            if status == E_OUTPUT_CAPACITY or (status in (DONE, NEED_MORE_INPUT) and res.written > self._U):
                status = E_EXTRANEOUS_IMAGE_DATA
            raise_for(status, (res.aux[0], res.aux[1]))
            self._state = state
            w, h = self.size
            now = min(res.written, self._U)
            if now > self._defiltered:
                desc = s.image_desc(None, self._d_rows, self._d_storage, w, h, self.depth, self.channels, self.interlaced,
                                    self.standard, rows_cap=self._d_rows.numel())
                desc.reserved = IMAGE_OVERDRAW if overdraw else 0
                ures = s.unfilter_resume(desc, self._d_work, self._defiltered, now)
                raise_for(ures.status, (ures.aux[0], ures.aux[1]))
                self.defiltered_total += ures.written
                self._defiltered = now
                self._stale = self._stale or ures.written > 0
            self._continue = status == NEED_MORE_INPUT

        def push_ancillary_iend(self):
            """push(ancillary: IEND) (:137-142)."""
            if self._continue:
                raise DecodingError(E_INCOMPLETE_DATASTREAM)
