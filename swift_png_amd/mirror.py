"""Python mirror of the reference's own API for the hot path, delegating to the C ABI.

Names, argument meaning and error behaviour follow the Swift originals so that the parity tests
read like the reference's tests.  State that the reference keeps as a resumable state machine
(LZ77.InflatorState, PNG.Decoder.row/pass) is kept here as "bytes pushed so far": the device path
decodes whole streams, so every push re-runs the stream from its start (results are identical by
construction; the streaming cost model is not -- see DESIGN.md "Out of scope").
"""
from __future__ import annotations

from . import (DONE, FORMAT_GZIP, FORMAT_IOS, FORMAT_ZLIB, NEED_MORE_INPUT, DecodingError,
               E_EXTRANEOUS_COMPRESSED_DATA, E_INCOMPLETE_DATASTREAM, E_OUTPUT_CAPACITY)

_DELAY_FORMATS = {1: (8, 1), 2: (8, 2), 3: (8, 3), 4: (8, 4), 6: (16, 3), 8: (16, 4)}


class LZ77:
    class Format:
        zlib = FORMAT_ZLIB
        ios = FORMAT_IOS

    class Inflator:
        """LZ77.Inflator (Sources/LZ77/Inflator/LZ77.Inflator.swift:8-62)."""

        def __init__(self, format=FORMAT_ZLIB, session=None):
            from . import load
            self._s = session or load()
            self._format = format
            self._in = bytearray()
            self._out = b""
            self._cursor = 0
            self._terminal = False

        def push(self, data) -> object:
            """Returns None once a complete stream has been received, () while it wants more."""
            from . import raise_for
            if self._terminal:
                return None                      # .terminal: remaining input is ignored (:38-40)
            self._in += bytes(data)
            cap = max(1 << 16, 1100 * len(self._in))
            while True:
                status, out, _, aux = self._s.inflate(bytes(self._in), self._format, cap)
                if status != E_OUTPUT_CAPACITY:
                    break
                cap *= 4                         # the reference's output buffer is unbounded
            raise_for(status, aux)
            self._out = out
            self._terminal = status == DONE
            return None if self._terminal else ()

        def pull(self, count=None):
            if count is None:                    # pull() -> everything available (:58-61)
                data, self._cursor = self._out[self._cursor:], len(self._out)
                return data
            if len(self._out) - self._cursor < count:
                return None                      # pull(_:) -> nil (:53-56)
            data = self._out[self._cursor:self._cursor + count]
            self._cursor += count
            return data


    class Deflator:
        """LZ77.Deflator (Sources/LZ77/Deflator/LZ77.Deflator.swift:8-44): init(format:level:exponent:hint:),
        push(_:last:), pull(), pop().  The device compresses whole streams (the concatenated output does not
        depend on how the input was pushed), so the bytes become available with the push
        that carries last: true; pop() then hands out complete chunks of 2 * hint bytes (the reference's chunk is
        2 * ManagedBuffer.capacity >= 2 * hint bytes, a platform-dependent size) and pull() also the final
        partial one -- their concatenation is the reference's stream bit for bit.  One known difference, on a
        reference bug: a row-by-row pushed stream whose last non-final compress() leaves fewer than three bytes
        queued makes the reference emit a corrupt stored tail (DeflatorBuffers.Stream.swift:45-60); this class
        emits the correct stream instead."""

        def __init__(self, format=FORMAT_ZLIB, level=9, exponent=15, hint=1 << 12, session=None):
            from . import load
            if not 8 <= exponent <= 15:
                raise ValueError("exponent must be 8 ... 15")
            self._s = session or load()
            self._format, self._level, self._exponent = format, level, exponent
            self._chunk = 2 * max(int(hint), 1)
            self._in = bytearray()
            self._out = None
            self._cursor = 0

        def push(self, data, last=False):
            if self._out is not None:
                raise RuntimeError("push after the last block")
            self._in += bytes(data)
            if last:
                self._out = self._s.deflate(bytes(self._in), self._level, self._format, self._exponent)
                self._in = bytearray()

        def pop(self):
            """A complete chunk, or None (:40-43)."""
            if self._out is None or len(self._out) - self._cursor < self._chunk:
                return None
            data = self._out[self._cursor:self._cursor + self._chunk]
            self._cursor += self._chunk
            return data

        def pull(self):
            """A complete chunk if there is one, else whatever has been written so far, else None (:31-35)."""
            data = self.pop()
            if data is not None:
                return data
            if self._out is None or self._cursor >= len(self._out):
                return None
            data, self._cursor = self._out[self._cursor:], len(self._out)
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
            from . import load, storage_size
            self._s = session or load()
            self.size, self.depth, self.channels = tuple(size), depth, channels
            self.interlaced, self.standard = bool(interlaced), standard
            self.storage = bytes(storage_size(size[0], size[1], depth, channels))
            self._idat = bytearray()
            self._continue = True                # PNG.Decoder.continue (:22-23)

        def push(self, data: bytes):
            """push(data:) (:88-102): one call per IDAT chunk."""
            from . import raise_for
            if not self._continue:
                raise DecodingError(E_EXTRANEOUS_COMPRESSED_DATA)     # PNG.Decoder.swift:51-55
            self._idat += bytes(data)
            w, h = self.size
            status, storage, aux = self._s.decode(bytes(self._idat), w, h, self.depth, self.channels,
                                                  self.interlaced, self.standard, self.storage)
            raise_for(status, aux)
            self.storage = storage
            self._continue = status == NEED_MORE_INPUT

        def push_ancillary_iend(self):
            """push(ancillary: IEND) (:137-142)."""
            if self._continue:
                raise DecodingError(E_INCOMPLETE_DATASTREAM)
