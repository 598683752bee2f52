// HipLZ4Batch.cs -- batch entry points of liblz4hip.so for C# callers (source only: no .NET toolchain in
// the build environment; the same calls are exercised through ctypes by lz4net_amd/stream.py,
// lz4net_amd/legacy_frame.py and LZ4Codec.WrapMany/UnwrapMany in lz4net_amd/codec.py).
//
// lz4net itself has no batch API: callers loop over LZ4Codec.Encode/Decode (LZ4Stream.FlushCurrentChunk /
// AcquireNextChunk, src/LZ4/LZ4Stream.cs:239-312).  A GPU pays off per batch, so this is the call a
// maintainer would route LZ4Stream's chunks or arrays of Wrap()ped messages through: all blocks of one call
// go to the device in one lz4hip_encode_batch_host / lz4hip_decode_batch_host (include/lz4hip.h).
using System;
using System.Runtime.InteropServices;

namespace LZ4hip
{
    public static class HipLZ4Batch
    {
        private const string Lib = "lz4hip";
        private const int ModeFast = 0, ModeHC = 1;                 // LZ4HIP_MODE_FAST / LZ4HIP_MODE_HC
        private const int ErrFirst = -2000000003, ErrLast = -2000000001;

        // struct lz4hip_batch (include/lz4hip.h), field for field
        [StructLayout(LayoutKind.Sequential)]
        private unsafe struct Batch
        {
            public byte* src; public long* src_off; public long src_stride; public int* src_len;
            public byte* dst; public long* dst_off; public long dst_stride; public int* dst_cap;
            public int dst_cap_all; public int src_len_all;
            public int* result; public long n_blocks;
        }

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        private static extern unsafe int lz4hip_encode_batch_host(Batch* b, int mode);

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        private static extern unsafe int lz4hip_decode_batch_host(Batch* b, int knownOutputSize);

        // Sharded over the GPUs of the node: block i -> the (i mod N)-th device of deviceMask (bit d = device d,
        // 0 = all visible devices); include/lz4hip.h, SURVEY.md 8e.  No launcher, no collective.
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        private static extern unsafe int lz4hip_encode_batch_host_multi(Batch* b, int mode, ulong deviceMask);

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        private static extern unsafe int lz4hip_decode_batch_host_multi(Batch* b, int knownOutputSize, ulong deviceMask);

        /// <summary>Devices the batch calls shard over (bit d = HIP device d; 0 = every visible device).</summary>
        public static ulong DeviceMask = 0;

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        private static extern int lz4hip_compressBound(int isize);

        private static void Check(int rc)
        {
            if (rc >= ErrFirst && rc <= ErrLast) throw new InvalidOperationException("liblz4hip failed: " + rc);
        }

        /// <summary>Compresses every block (LZ4Codec.Encode / EncodeHC semantics per block, outputLength =
        /// MaximumOutputLength).  results[i] is what Encode would have returned for block i.</summary>
        public static unsafe byte[][] Encode(byte[][] blocks, bool highCompression, out int[] results)
        {
            int n = blocks.Length;
            long total = 0, capTotal = 0;
            var srcOff = new long[n]; var dstOff = new long[n]; var lens = new int[n]; var caps = new int[n];
            for (int i = 0; i < n; i++)
            {
                srcOff[i] = total; lens[i] = blocks[i].Length; total += lens[i];
                dstOff[i] = capTotal; caps[i] = lz4hip_compressBound(lens[i]); capTotal += caps[i];
            }
            var src = new byte[Math.Max(total, 1)]; var dst = new byte[Math.Max(capTotal, 1)];
            for (int i = 0; i < n; i++) Buffer.BlockCopy(blocks[i], 0, src, (int)srcOff[i], lens[i]);
            results = new int[n];
            fixed (byte* ps = src, pd = dst)
            fixed (long* so = srcOff, dof = dstOff)
            fixed (int* sl = lens, dc = caps, res = results)
            {
                var b = new Batch { src = ps, src_off = so, src_len = sl, dst = pd, dst_off = dof, dst_cap = dc, result = res, n_blocks = n };
                Check(lz4hip_encode_batch_host_multi(&b, highCompression ? ModeHC : ModeFast, DeviceMask));
            }
            var output = new byte[n][];
            for (int i = 0; i < n; i++)
            {
                int len = results[i];
                if (highCompression && len <= 0) len = results[i] = -1;       // src/LZ4pn/LZ4Codec.Unsafe.cs:576-578
                output[i] = new byte[Math.Max(len, 0)];
                if (len > 0) Buffer.BlockCopy(dst, (int)dstOff[i], output[i], 0, len);
            }
            return output;
        }

        /// <summary>Decompresses every block into a buffer of outputLengths[i] bytes
        /// (LZ4Codec.Decode(..., knownOutputLength: true) semantics per block).</summary>
        public static unsafe byte[][] Decode(byte[][] blocks, int[] outputLengths)
        {
            int n = blocks.Length;
            long total = 0, outTotal = 0;
            var srcOff = new long[n]; var dstOff = new long[n]; var lens = new int[n];
            for (int i = 0; i < n; i++)
            {
                srcOff[i] = total; lens[i] = blocks[i].Length; total += lens[i];
                dstOff[i] = outTotal; outTotal += outputLengths[i];
            }
            var src = new byte[Math.Max(total, 1)]; var dst = new byte[Math.Max(outTotal, 1)];
            for (int i = 0; i < n; i++) Buffer.BlockCopy(blocks[i], 0, src, (int)srcOff[i], lens[i]);
            var results = new int[n];
            fixed (byte* ps = src, pd = dst)
            fixed (long* so = srcOff, dof = dstOff)
            fixed (int* sl = lens, dc = outputLengths, res = results)
            {
                var b = new Batch { src = ps, src_off = so, src_len = sl, dst = pd, dst_off = dof, dst_cap = dc, result = res, n_blocks = n };
                Check(lz4hip_decode_batch_host_multi(&b, 1, DeviceMask));
            }
            var output = new byte[n][];
            for (int i = 0; i < n; i++)
            {
                if (results[i] != lens[i])                                    // src/LZ4pn/LZ4Codec.Unsafe.cs:373-378
                    throw new ArgumentException("LZ4 block is corrupted, or invalid length has been given.");
                output[i] = new byte[outputLengths[i]];
                Buffer.BlockCopy(dst, (int)dstOff[i], output[i], 0, outputLengths[i]);
            }
            return output;
        }
    }
}
