// LZ4Codec.hpp -- C++ host-side mirror of lz4net's public static API `LZ4.LZ4Codec`
// (src/LZ4/LZ4Codec.cs:298-463) on top of the C ABI of liblz4hip.so (include/lz4hip.h).
//
// lz4net's host code is C#; this environment has no .NET toolchain, so the executable host mirrors are
// this header (C++) and lz4net_amd/codec.py (Python, used by the parity tests); the C# adapter itself is
// bindings/csharp/HipLZ4Service.cs.  All three implement the same call sequence:
//   LZ4Codec.Encode/EncodeHC/Decode (facade)  ->  HipLZ4Service (ILZ4Service, src/LZ4/ILZ4Service.cs:30-36)
//   ->  CheckArguments (src/LZ4ps/LZ4Codec.cs:151-170) + the L1 wrappers' result mapping
//       (src/LZ4pn/LZ4Codec.Unsafe.cs:307-326,366-439,559-606)  ->  lz4hip_* (the gfx950 kernels).
//
// .NET exceptions map to: ArgumentNullException / ArgumentException -> lz4net::ArgumentException (std::invalid_argument),
// InvalidOperationException -> lz4net::InvalidOperationException (std::runtime_error).
// Arrays are (pointer, length) pairs; a null pointer is C#'s null array.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../lz4hip.h"

namespace lz4net {

struct ArgumentException : std::invalid_argument { using std::invalid_argument::invalid_argument; };
struct ArgumentNullException : ArgumentException { using ArgumentException::ArgumentException; };
struct InvalidOperationException : std::runtime_error { using std::runtime_error::runtime_error; };

// a C# byte[]: pointer + Length
struct ByteArray {
    uint8_t* data; int length;
    ByteArray(uint8_t* d, int n) : data(d), length(n) {}
    ByteArray(std::vector<uint8_t>& v) : data(v.data()), length((int)v.size()) {}
    ByteArray(std::nullptr_t) : data(nullptr), length(0) {}
};

namespace detail {

inline int check(int rc)
{
    if (rc <= LZ4HIP_E_DEVICE && rc >= LZ4HIP_E_MEMORY)
        throw InvalidOperationException(std::string("liblz4hip failed: ") + lz4hip_last_error());
    return rc;
}

// LZ4Codec.CheckArguments, src/LZ4ps/LZ4Codec.cs:151-170 (same order of tests)
inline void check_arguments(const ByteArray& input, int inputOffset, int& inputLength,
                            const ByteArray& output, int outputOffset, int& outputLength)
{
    if (inputLength < 0) {
        if (!input.data) throw ArgumentNullException("input");
        inputLength = input.length - inputOffset;
    }
    if (inputLength == 0) { outputLength = 0; return; }
    if (!input.data) throw ArgumentNullException("input");
    if (inputOffset < 0 || inputOffset + inputLength > input.length)
        throw ArgumentException("inputOffset and inputLength are invalid for given input");
    if (outputLength < 0) {
        if (!output.data) throw ArgumentNullException("output");
        outputLength = output.length - outputOffset;
    }
    if (!output.data) throw ArgumentNullException("output");
    if (outputOffset < 0 || outputOffset + outputLength > output.length)
        throw ArgumentException("outputOffset and outputLength are invalid for given output");
}

}  // namespace detail

// ILZ4Service over liblz4hip.so (what bindings/csharp/HipLZ4Service.cs is in C#)
class HipLZ4Service {
public:
    std::string CodecName() const { return lz4hip_codec_name(); }

    int Encode(ByteArray input, int inputOffset, int inputLength, ByteArray output, int outputOffset, int outputLength) const
    {
        detail::check_arguments(input, inputOffset, inputLength, output, outputOffset, outputLength);
        if (outputLength == 0) return 0;
        return detail::check(lz4hip_compress_limitedOutput((const char*)input.data + inputOffset, (char*)output.data + outputOffset,
                                                           inputLength, outputLength));
    }
    int EncodeHC(ByteArray input, int inputOffset, int inputLength, ByteArray output, int outputOffset, int outputLength) const
    {
        detail::check_arguments(input, inputOffset, inputLength, output, outputOffset, outputLength);
        if (outputLength == 0) return 0;
        const int n = detail::check(lz4hip_compressHC_limitedOutput((const char*)input.data + inputOffset, (char*)output.data + outputOffset,
                                                                    inputLength, outputLength));
        return n <= 0 ? -1 : n;                               // src/LZ4pn/LZ4Codec.Unsafe.cs:576-578
    }
    int Decode(ByteArray input, int inputOffset, int inputLength, ByteArray output, int outputOffset, int outputLength,
               bool knownOutputLength) const
    {
        detail::check_arguments(input, inputOffset, inputLength, output, outputOffset, outputLength);
        if (outputLength == 0) return 0;
        if (knownOutputLength) {
            const int consumed = detail::check(lz4hip_uncompress_bounded((const char*)input.data + inputOffset, inputLength,
                                                                         (char*)output.data + outputOffset, outputLength));
            if (consumed != inputLength)                      // src/LZ4pn/LZ4Codec.Unsafe.cs:373-378
                throw ArgumentException("LZ4 block is corrupted, or invalid length has been given.");
            return outputLength;
        }
        const int produced = detail::check(lz4hip_uncompress_unknownOutputSize((const char*)input.data + inputOffset,
                                                                               (char*)output.data + outputOffset, inputLength, outputLength));
        if (produced < 0) throw ArgumentException("LZ4 block is corrupted, or invalid length has been given.");
        return produced;
    }
};

// static facade, src/LZ4/LZ4Codec.cs:298-463
class LZ4Codec {
    static const HipLZ4Service& service() { static HipLZ4Service s; return s; }

public:
    static std::string CodecName()
    {
        const std::string n = service().CodecName();
        return n + "/" + n + "/" + n + "HC";                 // "{Encoder}/{Decoder}/{EncoderHC}HC", :298-308
    }
    static int MaximumOutputLength(int inputLength) { return inputLength + inputLength / 255 + 16; }   // :313-316

    static int Encode(ByteArray input, int inputOffset, int inputLength, ByteArray output, int outputOffset, int outputLength)
    { return service().Encode(input, inputOffset, inputLength, output, outputOffset, outputLength); }
    static int EncodeHC(ByteArray input, int inputOffset, int inputLength, ByteArray output, int outputOffset, int outputLength)
    { return service().EncodeHC(input, inputOffset, inputLength, output, outputOffset, outputLength); }
    static int Decode(ByteArray input, int inputOffset, int inputLength, ByteArray output, int outputOffset,
                      int outputLength = 0, bool knownOutputLength = false)
    { return service().Decode(input, inputOffset, inputLength, output, outputOffset, outputLength, knownOutputLength); }

    // allocating overloads: Encode64(byte[], int, int) / Decode64(byte[], int, int, int), src/LZ4pn/LZ4Codec.Unsafe.cs:335-353,427-439
    static std::vector<uint8_t> Encode(ByteArray input, int inputOffset, int inputLength, bool hc = false)
    {
        if (inputLength < 0) { if (!input.data) throw ArgumentNullException("input"); inputLength = input.length - inputOffset; }
        if (!input.data) throw ArgumentNullException("input");
        if (inputOffset < 0 || inputOffset + inputLength > input.length)
            throw ArgumentException("inputOffset and inputLength are invalid for given input");
        std::vector<uint8_t> result((size_t)MaximumOutputLength(inputLength));
        const int length = hc ? EncodeHC(input, inputOffset, inputLength, result, 0, (int)result.size())
                              : Encode(input, inputOffset, inputLength, result, 0, (int)result.size());
        if (length < 0) throw InvalidOperationException("Compression has been corrupted");
        result.resize((size_t)length);
        return result;
    }
    static std::vector<uint8_t> EncodeHC(ByteArray input, int inputOffset, int inputLength) { return Encode(input, inputOffset, inputLength, true); }
    static std::vector<uint8_t> Decode(ByteArray input, int inputOffset, int inputLength, int outputLength)
    {
        if (inputLength < 0) { if (!input.data) throw ArgumentNullException("input"); inputLength = input.length - inputOffset; }
        if (!input.data) throw ArgumentNullException("input");
        if (inputOffset < 0 || inputOffset + inputLength > input.length)
            throw ArgumentException("inputOffset and inputLength are invalid for given input");
        std::vector<uint8_t> result((size_t)(outputLength > 0 ? outputLength : 0));
        uint8_t dummy = 0;
        ByteArray out(result.empty() ? &dummy : result.data(), (int)result.size());
        const int length = Decode(input, inputOffset, inputLength, out, 0, outputLength, true);
        if (length != outputLength) throw ArgumentException("outputLength is not valid");
        return result;
    }
};

}  // namespace lz4net
