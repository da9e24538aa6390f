// Test driver for CellShape::loadFromFile and ImageLibraryFile::loadFromFile of the C++ host mirror.  Per .mil file given:
// one JSON line with image size, count, the names and an FNV-1a of the BGR bytes.  Per .mcs file given, one JSON
// line with what tests/golden/cells_mcs.json records (name, size, active pixels, FNV-1a of the
// thresholded mask, tiling parameters, and the same after resized(64)).
#include <cstdio>
#include <string>
#include <vector>

#include "../mosaicmagnifique_amd/csrc/host/HIPPhotomosaicGenerator.h"

static void params(const CellShape& s)
{
    printf("{\"rowSpacing\": %d, \"colSpacing\": %d, \"alternateRowSpacing\": %d, \"alternateColSpacing\": %d, "
           "\"alternateRowOffset\": %d, \"alternateColOffset\": %d, \"alternateColFlipHorizontal\": %s, "
           "\"alternateColFlipVertical\": %s, \"alternateRowFlipHorizontal\": %s, \"alternateRowFlipVertical\": %s}",
           s.rowSpacing, s.colSpacing, s.alternateRowSpacing, s.alternateColSpacing, s.alternateRowOffset, s.alternateColOffset,
           s.alternateColFlipHorizontal ? "true" : "false", s.alternateColFlipVertical ? "true" : "false",
           s.alternateRowFlipHorizontal ? "true" : "false", s.alternateRowFlipVertical ? "true" : "false");
}

// --bounds FILE: "nb nq" then nb bound rects and nq query rects ("x y w h" each) -> one line of 0 / 1 per query:
// GridUtility::BoundsUnion::meets (the in-bounds test of findCellState, GridGenerator.cpp:124-140, on the union of the bounds)
static int bounds_mode(const char* path)
{
    using GridUtility::Rect;
    FILE* f = fopen(path, "r");
    if (!f) return 2;
    int nb = 0, nq = 0;
    if (fscanf(f, "%d %d", &nb, &nq) != 2) { fclose(f); return 2; }
    std::vector<Rect> bounds((size_t)nb), queries((size_t)nq);
    for (Rect& r : bounds) if (fscanf(f, "%d %d %d %d", &r.x, &r.y, &r.width, &r.height) != 4) { fclose(f); return 2; }
    for (Rect& r : queries) if (fscanf(f, "%d %d %d %d", &r.x, &r.y, &r.width, &r.height) != 4) { fclose(f); return 2; }
    fclose(f);
    const GridUtility::BoundsUnion u(bounds);
    for (const Rect& q : queries) putchar(u.meets(q) ? '1' : '0');
    putchar('\n');
    return 0;
}

int main(int argc, char** argv)
{
    if (argc == 3 && std::string(argv[1]) == "--bounds") return bounds_mode(argv[2]);
    for (int i = 1; i < argc; ++i) {
        const std::string path = argv[i];
        if (path.size() > 4 && path.compare(path.size() - 4, 4, ".mil") == 0) {
            try {
                const ImageLibraryFile lib = ImageLibraryFile::loadFromFile(path);
                unsigned long long fnv = 1469598103934665603ull;
                for (uint8_t v : lib.bgr) fnv = (fnv ^ v) * 1099511628211ull;
                printf("{\"imageSize\": %d, \"count\": %zu, \"bytes\": %zu, \"fnv\": \"%016llx\", \"names\": [", lib.imageSize, lib.count(), lib.bgr.size(), fnv);
                for (size_t k = 0; k < lib.names.size(); ++k) {
                    printf("%s\"", k ? ", " : "");
                    for (unsigned char ch : lib.names[k]) {          // (test names hold no quotes or backslashes; control bytes escaped)
                        if (ch < 0x20 || ch == '"' || ch == '\\') printf("\\u%04x", ch); else putchar(ch);
                    }
                    printf("\"");
                }
                printf("]}\n");
            } catch (const std::exception& e) {
                printf("{\"error\": \"%s\"}\n", e.what());
            }
            continue;
        }
        try {
            const CellShape shape = CellShape::loadFromFile(argv[i]);
            const auto& mask = shape.getCellMask(false, false);
            unsigned long long fnv = 1469598103934665603ull;
            long active = 0;
            for (uint8_t v : mask) { const int on = v > 127; active += on; fnv = (fnv ^ (unsigned)on) * 1099511628211ull; }
            const CellShape small = shape.getSize() >= 64 ? shape.resized(64) : shape;     // no enlargement (INTER_CUBIC) in the hot path
            long small_active = 0;
            for (uint8_t v : small.getCellMask(false, false)) small_active += v != 0;
            // flipped variants: the (1,1) mask read backwards must be the (0,0) mask
            const auto& both = shape.getCellMask(true, true);
            bool flips_ok = both.size() == mask.size();
            for (size_t k = 0; flips_ok && k < mask.size(); ++k) flips_ok = both[mask.size() - 1 - k] == mask[k];
            printf("{\"name\": \"%s\", \"size\": %d, \"active\": %ld, \"mask_fnv\": \"%016llx\", \"flips_ok\": %s, \"params\": ",
                   shape.name.c_str(), shape.getSize(), active, fnv, flips_ok ? "true" : "false");
            params(shape);
            printf(", \"resized64\": {\"active\": %ld, \"params\": ", small_active);
            params(small);
            printf("}}\n");
        } catch (const std::exception& e) {
            printf("{\"error\": \"%s\"}\n", e.what());
        }
    }
    return 0;
}
