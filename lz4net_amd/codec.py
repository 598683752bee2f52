"""Host-side mirror of lz4net's public static API ``LZ4.LZ4Codec`` (src/LZ4/LZ4Codec.cs:298-599) on top
of the C ABI of liblz4hip.so -- same method names, argument meaning, defaults, return values and
error behaviour, so the parity tests read like the reference's own tests.

The layering mirrors the reference:
  LZ4Codec.Encode/EncodeHC/Decode (facade, src/LZ4/LZ4Codec.cs:328-463)
    -> HipLZ4Service (this module; stands where Unsafe64LZ4Service sits, src/LZ4/Services/Unsafe64LZ4Service.cs:30-55)
      -> argument checks + exception mapping of the L1 wrappers (src/LZ4pn/LZ4Codec.Unsafe.cs:307-439,559-606,
         src/LZ4ps/LZ4Codec.cs:151-170)
        -> lz4hip_compress_limitedOutput / lz4hip_compressHC_limitedOutput / lz4hip_uncompress_bounded /
           lz4hip_uncompress_unknownOutputSize (include/lz4hip.h), i.e. the gfx950 kernels.

.NET exceptions are mapped to Python ones with the same names (subclasses of ValueError / RuntimeError).
The C# source of the real shim is bindings/csharp/HipLZ4Service.cs (cannot be compiled here: no .NET).
"""
from __future__ import annotations

import numpy as np

from . import _lib


class ArgumentException(ValueError):
    pass


class ArgumentNullException(ArgumentException):
    pass


class InvalidOperationException(RuntimeError):
    pass


_CORRUPT = "LZ4 block is corrupted, or invalid length has been given."
_INT_MAX = 2 ** 31 - 1


def _as_bytes(buf, name):
    if buf is None:
        raise ArgumentNullException(name)
    if isinstance(buf, np.ndarray):
        if buf.dtype != np.uint8 or buf.ndim != 1 or not buf.flags.c_contiguous:
            raise ArgumentException(f"{name} must be a contiguous 1-D uint8 array")
        return buf
    return np.frombuffer(buf, dtype=np.uint8)   # bytes / bytearray / memoryview (zero-copy)


def _check_arguments(inp, input_offset, input_length, out, output_offset, output_length):
    """LZ4Codec.CheckArguments (src/LZ4ps/LZ4Codec.cs:151-170), including its order of tests."""
    if input_length < 0:
        if inp is None:
            raise ArgumentNullException("input")
        input_length = len(inp) - input_offset
    if input_length == 0:
        return input_length, 0
    if inp is None:
        raise ArgumentNullException("input")
    if input_offset < 0 or input_offset + input_length > len(inp):
        raise ArgumentException("inputOffset and inputLength are invalid for given input")
    if output_length < 0:
        if out is None:
            raise ArgumentNullException("output")
        output_length = len(out) - output_offset
    if out is None:
        raise ArgumentNullException("output")
    if output_offset < 0 or output_offset + output_length > len(out):
        raise ArgumentException("outputOffset and outputLength are invalid for given output")
    return input_length, output_length


class HipLZ4Service:
    """ILZ4Service (src/LZ4/ILZ4Service.cs:30-36) implemented on liblz4hip.so."""

    @property
    def CodecName(self) -> str:
        return _lib.lib().lz4hip_codec_name().decode()

    @staticmethod
    def _ptr(a: np.ndarray, offset: int) -> int:
        return a.ctypes.data + offset

    def Encode(self, inp, input_offset, input_length, out, output_offset, output_length) -> int:
        # Encode64(array form), src/LZ4pn/LZ4Codec.Unsafe.cs:307-326
        input_length, output_length = _check_arguments(inp, input_offset, input_length, out, output_offset, output_length)
        if output_length == 0:
            return 0
        return _lib.check(_lib.lib().lz4hip_compress_limitedOutput(
            self._ptr(inp, input_offset), self._ptr(out, output_offset), input_length, output_length))

    def EncodeHC(self, inp, input_offset, input_length, out, output_offset, output_length) -> int:
        # Encode64HC(array form), src/LZ4pn/LZ4Codec.Unsafe.cs:559-580: <= 0 from the core becomes -1
        input_length, output_length = _check_arguments(inp, input_offset, input_length, out, output_offset, output_length)
        if output_length == 0:
            return 0
        length = _lib.check(_lib.lib().lz4hip_compressHC_limitedOutput(
            self._ptr(inp, input_offset), self._ptr(out, output_offset), input_length, output_length))
        return -1 if length <= 0 else length

    def Decode(self, inp, input_offset, input_length, out, output_offset, output_length, known_output_length) -> int:
        # Decode64, src/LZ4pn/LZ4Codec.Unsafe.cs:366-418
        input_length, output_length = _check_arguments(inp, input_offset, input_length, out, output_offset, output_length)
        if output_length == 0:
            return 0
        if known_output_length:
            length = _lib.check(_lib.lib().lz4hip_uncompress_bounded(
                self._ptr(inp, input_offset), input_length, self._ptr(out, output_offset), output_length))
            if length != input_length:
                raise ArgumentException(_CORRUPT)
            return output_length
        length = _lib.check(_lib.lib().lz4hip_uncompress_unknownOutputSize(
            self._ptr(inp, input_offset), self._ptr(out, output_offset), input_length, output_length))
        if length < 0:
            raise ArgumentException(_CORRUPT)
        return length


class _LZ4CodecMeta(type):
    @property
    def CodecName(cls) -> str:
        # "{Encoder}/{Decoder}/{EncoderHC}HC", src/LZ4/LZ4Codec.cs:298-308
        n = cls._service.CodecName
        return f"{n}/{n}/{n}HC"


class LZ4Codec(metaclass=_LZ4CodecMeta):
    """Static API of LZ4.LZ4Codec.  Buffers are bytes / bytearray / 1-D uint8 numpy arrays."""

    _service = HipLZ4Service()

    @staticmethod
    def MaximumOutputLength(input_length: int) -> int:
        return input_length + input_length // 255 + 16            # src/LZ4/LZ4Codec.cs:313-316

    # ---- Encode / EncodeHC: src/LZ4/LZ4Codec.cs:328-399 -------------------------------------------
    @classmethod
    def _encode(cls, hc, inp, input_offset, input_length, out, output_offset, output_length):
        fn = cls._service.EncodeHC if hc else cls._service.Encode
        if out is not None:                                        # 6-argument overload
            inp_a = None if inp is None else _as_bytes(inp, "input")
            out_a = _as_bytes(out, "output")
            if not out_a.flags.writeable:
                raise ArgumentException("output is read-only")
            return fn(inp_a, input_offset, input_length, out_a, output_offset, output_length)
        # allocating overload: Encode64(byte[], int, int), src/LZ4pn/LZ4Codec.Unsafe.cs:335-353
        if inp is None:
            raise ArgumentNullException("input")
        inp_a = _as_bytes(inp, "input")
        if input_length < 0:
            input_length = len(inp_a) - input_offset
        if input_offset < 0 or input_offset + input_length > len(inp_a):
            raise ArgumentException("inputOffset and inputLength are invalid for given input")
        result = np.zeros(cls.MaximumOutputLength(input_length), dtype=np.uint8)
        length = fn(inp_a, input_offset, input_length, result, 0, len(result))
        if length < 0:
            raise InvalidOperationException("Compression has been corrupted")
        return bytes(result[:length])

    @classmethod
    def Encode(cls, input, inputOffset=0, inputLength=-1, output=None, outputOffset=0, outputLength=-1):
        return cls._encode(False, input, inputOffset, inputLength, output, outputOffset, outputLength)

    @classmethod
    def EncodeHC(cls, input, inputOffset=0, inputLength=-1, output=None, outputOffset=0, outputLength=-1):
        return cls._encode(True, input, inputOffset, inputLength, output, outputOffset, outputLength)

    # ---- Decode: src/LZ4/LZ4Codec.cs:430-463 --------------------------------------------------------
    @classmethod
    def Decode(cls, input, inputOffset=0, inputLength=-1, output=None, outputOffset=0, outputLength=0,
               knownOutputLength=False):
        """Decode(input, inputOffset, inputLength, output, outputOffset, outputLength=0, knownOutputLength=false)
        -> bytes written, or the allocating overload Decode(input, inputOffset, inputLength, outputLength) -> bytes
        (4th positional argument an int, or output=None with outputLength given)."""
        if isinstance(output, int):
            output, outputLength = None, output
        if output is not None:
            inp_a = None if input is None else _as_bytes(input, "input")
            out_a = _as_bytes(output, "output")
            if not out_a.flags.writeable:
                raise ArgumentException("output is read-only")
            return cls._service.Decode(inp_a, inputOffset, inputLength, out_a, outputOffset, outputLength, knownOutputLength)
        # allocating overload: Decode64(byte[], int, int, int outputLength), src/LZ4pn/LZ4Codec.Unsafe.cs:427-439
        if input is None:
            raise ArgumentNullException("input")
        inp_a = _as_bytes(input, "input")
        if inputLength < 0:
            inputLength = len(inp_a) - inputOffset
        if inputOffset < 0 or inputOffset + inputLength > len(inp_a):
            raise ArgumentException("inputOffset and inputLength are invalid for given input")
        result = np.zeros(max(outputLength, 0), dtype=np.uint8)
        length = cls._service.Decode(inp_a, inputOffset, inputLength, result, 0, outputLength, True)
        if length != outputLength:
            raise ArgumentException("outputLength is not valid")
        return bytes(result)

    # ---- Wrap / WrapHC / Unwrap: src/LZ4/LZ4Codec.cs:471-599 ----------------------------------------
    @classmethod
    def _wrap(cls, input_buffer, input_offset, input_length, hc) -> bytes:
        buf = _as_bytes(input_buffer, "inputBuffer")
        input_length = min(len(buf) - input_offset, input_length)
        if input_length < 0:
            raise ArgumentException("inputBuffer size of inputLength is invalid")
        if input_length == 0:
            return bytes(8)
        out = np.zeros(input_length, dtype=np.uint8)               # outputLength = inputLength (not MaximumOutputLength)
        fn = cls._service.EncodeHC if hc else cls._service.Encode
        n = fn(buf, input_offset, input_length, out, 0, input_length)
        if n >= input_length or n <= 0:                             # stored raw (:527-533)
            return (int(input_length).to_bytes(4, "little") * 2) + bytes(buf[input_offset:input_offset + input_length])
        return int(input_length).to_bytes(4, "little") + int(n).to_bytes(4, "little") + bytes(out[:n])

    @classmethod
    def Wrap(cls, inputBuffer, inputOffset=0, inputLength=_INT_MAX) -> bytes:
        return cls._wrap(inputBuffer, inputOffset, inputLength, False)

    @classmethod
    def WrapHC(cls, inputBuffer, inputOffset=0, inputLength=_INT_MAX) -> bytes:
        return cls._wrap(inputBuffer, inputOffset, inputLength, True)

    @classmethod
    def Unwrap(cls, inputBuffer, inputOffset=0) -> bytes:
        buf = _as_bytes(inputBuffer, "inputBuffer")
        input_length = len(buf) - inputOffset
        if input_length < 8:
            raise ArgumentException("inputBuffer size is invalid")
        output_length = int.from_bytes(bytes(buf[inputOffset:inputOffset + 4]), "little", signed=True)
        input_length = int.from_bytes(bytes(buf[inputOffset + 4:inputOffset + 8]), "little", signed=True)
        if input_length > len(buf) - inputOffset - 8:
            raise ArgumentException("inputBuffer size is invalid or has been corrupted")
        if input_length >= output_length:
            return bytes(buf[inputOffset + 8:inputOffset + 8 + input_length])
        result = np.zeros(output_length, dtype=np.uint8)
        cls._service.Decode(buf, inputOffset + 8, input_length, result, 0, output_length, True)
        return bytes(result)

    # ---- batched Wrap / Unwrap: many messages, ONE GPU batch ---------------------------
    @classmethod
    def WrapMany(cls, messages, high_compression: bool = False) -> list:
        """[Wrap(m) for m in messages] with every message compressed in one lz4hip_encode_batch_host call."""
        import ctypes as C
        from . import _lib
        from .stream import _batch
        bufs = [bytes(_as_bytes(m, "inputBuffer")) for m in messages]
        idx = [i for i, m in enumerate(bufs) if len(m) > 0]
        out = [bytes(8)] * len(bufs)                                 # empty message -> 8 zero bytes (:497-498)
        if not idx:
            return out
        lens = np.array([len(bufs[i]) for i in idx], dtype=np.int32)
        offs = np.concatenate(([0], np.cumsum(lens[:-1], dtype=np.int64))).astype(np.int64)
        raw = np.frombuffer(b"".join(bufs[i] for i in idx), dtype=np.uint8)
        comp = np.zeros(raw.size, dtype=np.uint8)                    # outputLength = inputLength per message
        res = np.zeros(len(idx), dtype=np.int32)
        b = _batch(raw, offs, lens, comp, offs, lens, res)
        _lib.check(_lib.lib().lz4hip_encode_batch_host(C.byref(b), _lib.MODE_HC if high_compression else _lib.MODE_FAST))
        for j, i in enumerate(idx):
            ln, n, o = int(lens[j]), int(res[j]), int(offs[j])
            if n >= ln or n <= 0:                                    # stored raw (:527-533)
                out[i] = ln.to_bytes(4, "little") * 2 + bufs[i]
            else:
                out[i] = ln.to_bytes(4, "little") + n.to_bytes(4, "little") + comp[o:o + n].tobytes()
        return out

    @classmethod
    def UnwrapMany(cls, wrapped) -> list:
        """[Unwrap(w) for w in wrapped] with every compressed message decoded in one lz4hip_decode_batch_host call."""
        import ctypes as C
        from . import _lib
        from .stream import _batch
        bufs = [bytes(_as_bytes(w, "inputBuffer")) for w in wrapped]
        out, todo = [None] * len(bufs), []
        for i, w in enumerate(bufs):
            if len(w) < 8:
                raise ArgumentException("inputBuffer size is invalid")
            olen = int.from_bytes(w[0:4], "little", signed=True)
            ilen = int.from_bytes(w[4:8], "little", signed=True)
            if ilen > len(w) - 8:
                raise ArgumentException("inputBuffer size is invalid or has been corrupted")
            if ilen >= olen:
                out[i] = w[8:8 + ilen]
            else:
                todo.append((i, olen, ilen))
        if todo:
            src = np.frombuffer(b"".join(bufs[i][8:8 + il] for i, _, il in todo), dtype=np.uint8)
            src_len = np.array([il for _, _, il in todo], dtype=np.int32)
            dst_len = np.array([ol for _, ol, _ in todo], dtype=np.int32)
            src_off = np.concatenate(([0], np.cumsum(src_len[:-1], dtype=np.int64))).astype(np.int64)
            dst_off = np.concatenate(([0], np.cumsum(dst_len[:-1], dtype=np.int64))).astype(np.int64)
            dst = np.zeros(int(dst_len.astype(np.int64).sum()), dtype=np.uint8)
            res = np.zeros(len(todo), dtype=np.int32)
            b = _batch(src, src_off, src_len, dst, dst_off, dst_len, res)
            _lib.check(_lib.lib().lz4hip_decode_batch_host(C.byref(b), 1))
            if not (res == src_len).all():                           # Decode64: consumed != inputLength (Unsafe.cs:373-378)
                raise ArgumentException("LZ4 block is corrupted, or invalid length has been given.")
            for j, (i, ol, _) in enumerate(todo):
                out[i] = dst[int(dst_off[j]):int(dst_off[j]) + ol].tobytes()
        return out

