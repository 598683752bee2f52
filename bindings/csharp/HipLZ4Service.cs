// HipLZ4Service.cs -- the reference-side binding a lz4net maintainer would add to route
// LZ4Codec.Encode / EncodeHC / Decode through liblz4hip.so (include/lz4hip.h).
//
// SOURCE ONLY: this build container and the GPU box have no .NET toolchain (no dotnet/mono/csc), so
// this file cannot be compiled or exercised here; lz4net_amd/codec.py is the executable mirror of
// exactly this logic and is what the parity tests drive.  See INTEGRATION.md for where it plugs in.
//
// It sits where the other ILZ4Service adapters sit (src/LZ4/Services/Unsafe64LZ4Service.cs:30-55) and
// reproduces the L1 wrapper behaviour of LZ4pn (src/LZ4pn/LZ4Codec.Unsafe.cs:307-326,366-418,559-580):
// CheckArguments, "inputLength == 0 => 0", HC "<= 0 => -1", and the ArgumentException on corrupt input.

using System;
using System.Runtime.InteropServices;

namespace LZ4.Services
{
    internal class HipLZ4Service : ILZ4Service
    {
        private const string Lib = "lz4hip";   // liblz4hip.so / lz4hip.dll on the loader path

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        private static extern IntPtr lz4hip_codec_name();

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        private static extern int lz4hip_device_count();

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        private static extern unsafe int lz4hip_compress_limitedOutput(byte* source, byte* dest, int isize, int maxOutputSize);

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        private static extern unsafe int lz4hip_compressHC_limitedOutput(byte* source, byte* dest, int isize, int maxOutputSize);

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        private static extern unsafe int lz4hip_uncompress_bounded(byte* source, int isize, byte* dest, int osize);

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        private static extern unsafe int lz4hip_uncompress_unknownOutputSize(byte* source, byte* dest, int isize, int maxOutputSize);

        private const int LZ4HIP_E_FIRST = -2000000003, LZ4HIP_E_LAST = -2000000001;

        public HipLZ4Service()
        {
            // TryService<T>() (src/LZ4/LZ4Codec.cs:278-290) swallows this and leaves the service null,
            // exactly as it does when the mixed-mode assemblies fail to load.
            if (lz4hip_device_count() < 1) throw new NotSupportedException("no gfx950 device");
        }

        public string CodecName
        {
            get { return Marshal.PtrToStringAnsi(lz4hip_codec_name()); }
        }

        private static int Check(int result)
        {
            if (result >= LZ4HIP_E_FIRST && result <= LZ4HIP_E_LAST)
                throw new InvalidOperationException("liblz4hip failed: " + result);
            return result;
        }

        public unsafe int Encode(byte[] input, int inputOffset, int inputLength, byte[] output, int outputOffset, int outputLength)
        {
            LZ4ps.LZ4Codec.CheckArguments(input, inputOffset, ref inputLength, output, outputOffset, ref outputLength);
            if (outputLength == 0) return 0;
            fixed (byte* i = &input[inputOffset])
            fixed (byte* o = &output[outputOffset])
                return Check(lz4hip_compress_limitedOutput(i, o, inputLength, outputLength));
        }

        public unsafe int EncodeHC(byte[] input, int inputOffset, int inputLength, byte[] output, int outputOffset, int outputLength)
        {
            LZ4ps.LZ4Codec.CheckArguments(input, inputOffset, ref inputLength, output, outputOffset, ref outputLength);
            if (outputLength == 0) return 0;
            fixed (byte* i = &input[inputOffset])
            fixed (byte* o = &output[outputOffset])
            {
                var length = Check(lz4hip_compressHC_limitedOutput(i, o, inputLength, outputLength));
                return length <= 0 ? -1 : length;          // src/LZ4pn/LZ4Codec.Unsafe.cs:576-578
            }
        }

        public unsafe int Decode(byte[] input, int inputOffset, int inputLength, byte[] output, int outputOffset, int outputLength, bool knownOutputLength)
        {
            LZ4ps.LZ4Codec.CheckArguments(input, inputOffset, ref inputLength, output, outputOffset, ref outputLength);
            if (outputLength == 0) return 0;
            fixed (byte* i = &input[inputOffset])
            fixed (byte* o = &output[outputOffset])
            {
                if (knownOutputLength)
                {
                    var consumed = Check(lz4hip_uncompress_bounded(i, inputLength, o, outputLength));
                    if (consumed != inputLength)            // src/LZ4pn/LZ4Codec.Unsafe.cs:373-378
                        throw new ArgumentException("LZ4 block is corrupted, or invalid length has been given.");
                    return outputLength;
                }
                var produced = Check(lz4hip_uncompress_unknownOutputSize(i, o, inputLength, outputLength));
                if (produced < 0)                           // src/LZ4pn/LZ4Codec.Unsafe.cs:381-385
                    throw new ArgumentException("LZ4 block is corrupted, or invalid length has been given.");
                return produced;
            }
        }
    }
}

// ---- registration (patch to src/LZ4/LZ4Codec.cs, shown as a comment because it edits reference code) ----
//
//   static LZ4Codec()                                      // src/LZ4/LZ4Codec.cs:76-101
//   {
//       ...
//       InitializeLZ4hip();                                // new, before the others
//       InitializeLZ4mm(); InitializeLZ4cc(); InitializeLZ4n(); InitializeLZ4s();
//       ...
//   }
//   private static ILZ4Service _service_HIP;
//   private static void InitializeLZ4hip() { _service_HIP = TryService<HipLZ4Service>(); }   // runs AutoTest (:173-239)
//
//   SelectCodec (src/LZ4/LZ4Codec.cs:103-168): put _service_HIP first in the encoder, decoder and
//   encoderHC priority lists.  For single small blocks a maintainer may prefer to keep N64 first and
//   use HipLZ4Service only from batch-aware callers (LZ4Stream, Wrap over many payloads): one block per
//   call pays a PCIe round trip, the GPU pays off on batches (see INTEGRATION.md "Batches").
