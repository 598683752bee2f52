// C++ host mirror self test: the reference's load-time AutoTest (src/LZ4/LZ4Codec.cs:173-239) --
// Lorem x5 through Encode -> Decode(known) -> Decode(unknown), fast and HC -- plus the argument checks.
// `--no-gpu`: only the host logic (argument checks, MaximumOutputLength, loud failure without a device).
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/lz4net/LZ4Codec.hpp"

using namespace lz4net;

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL line %d: %s\n", __LINE__, #cond); fails++; } } while (0)
template <class E, class F> static bool throws(F f) { try { f(); } catch (const E&) { return true; } catch (...) { return false; } return false; }

int main(int argc, char** argv)
{
    const bool gpu = !(argc > 1 && std::strcmp(argv[1], "--no-gpu") == 0);
    std::string lorem = "Lorem ipsum dolor sit amet, consectetur adipisicing elit, sed do eiusmod tempor incididunt ut "
                        "labore et dolore magna aliqua. Ut enim ad minim veniam, quis nostrud exercitation ullamco laboris "
                        "nisi ut aliquip ex ea commodo consequat. Duis aute irure dolor in reprehenderit in voluptate velit "
                        "esse cillum dolore eu fugiat nulla pariatur. Excepteur sint occaecat cupidatat non proident, sunt "
                        "in culpa qui officia deserunt mollit anim id est laborum.";
    std::string text;
    for (int i = 0; i < 5; i++) text += lorem;
    std::vector<uint8_t> original(text.begin(), text.end());

    CHECK(LZ4Codec::MaximumOutputLength(65536) == 65809);
    std::vector<uint8_t> small(4), three = { 'a', 'b', 'c' };
    CHECK(LZ4Codec::Encode(ByteArray(nullptr), 0, 0, small, 0, 4) == 0);                       // inputLength == 0 => 0
    CHECK(throws<ArgumentNullException>([&] { LZ4Codec::Encode(ByteArray(nullptr), 0, 3, small, 0, 4); }));
    CHECK(throws<ArgumentException>([&] { LZ4Codec::Encode(three, 1, 3, small, 0, 4); }));
    CHECK(throws<ArgumentException>([&] { LZ4Codec::Decode(three, 0, 3, small, 2, 4, true); }));
    if (!gpu) {
        if (lz4hip_device_count() == 0)
            CHECK(throws<InvalidOperationException>([&] { LZ4Codec::Encode(original, 0, (int)original.size()); }));   // no CPU fallback
        std::printf(fails ? "host-logic self test FAILED\n" : "host-logic self test ok\n");
        return fails ? 1 : 0;
    }
    for (int hc = 0; hc < 2; hc++) {
        std::vector<uint8_t> comp = hc ? LZ4Codec::EncodeHC(original, 0, (int)original.size()) : LZ4Codec::Encode(original, 0, (int)original.size());
        CHECK(comp.size() > 0 && comp.size() < original.size());
        std::vector<uint8_t> back(original.size());
        CHECK(LZ4Codec::Decode(comp, 0, (int)comp.size(), back, 0, (int)back.size(), true) == (int)original.size());
        CHECK(back == original);
        std::vector<uint8_t> roomy(original.size() + 50);
        CHECK(LZ4Codec::Decode(comp, 0, (int)comp.size(), roomy, 0, (int)roomy.size(), false) == (int)original.size());
        CHECK(std::memcmp(roomy.data(), original.data(), original.size()) == 0);
        CHECK(LZ4Codec::Decode(comp, 0, (int)comp.size(), (int)original.size()) == original);
        CHECK(throws<ArgumentException>([&] { LZ4Codec::Decode(comp, 0, (int)comp.size() - 1, back, 0, (int)back.size(), true); }));
        std::vector<uint8_t> tiny(10);
        if (hc) CHECK(LZ4Codec::EncodeHC(original, 0, (int)original.size(), tiny, 0, 10) == -1);
        else    CHECK(LZ4Codec::Encode(original, 0, (int)original.size(), tiny, 0, 10) == 0);
        std::printf("%s: %zu -> %zu bytes\n", hc ? "EncodeHC" : "Encode", original.size(), comp.size());
    }
    std::printf("codec: %s\n", LZ4Codec::CodecName().c_str());
    std::printf(fails ? "self test FAILED\n" : "self test ok\n");
    return fails ? 1 : 0;
}
