// Test driver for the C++ host mirror: reads a scenario written by pytest, runs
// HIPPhotomosaicGenerator the way the reference's generator tests drive theirs
// (test/tst_CUDAGenerator.h:20-41,163-212: CreateCellGroup(size, steps, detail), setters,
// GridGenerator::getGridState, generateBestFits), writes the best-fit grids and the mosaic.
//   scenario.bin: int32 header {rows, cols, size, L, type, range, addition, 6 spacings/offsets,
//                 4 flips, detail percent, size steps, (inputs are u8) | colour scheme << 1}
//                 u8 inputs:  u8 BGR image, u8 BGR library (L x size x size x 3), u8 mask (size*size)
//                 f32 inputs: f32 image, f32 library, u8 mask, then the u8 BGR library for buildPhotomosaic
//   result.bin:   int32 steps, per step int32 gridH, gridW + int64 grid (-1 = nullopt), then the BGRA mosaic
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../mosaicmagnifique_amd/csrc/host/HIPPhotomosaicGenerator.h"

template <typename T>
static bool readAll(FILE* f, std::vector<T>& v) { return fread(v.data(), sizeof(T), v.size(), f) == v.size(); }

int main(int argc, char** argv)
{
    if (argc != 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[20];
    if (fread(h, sizeof h, 1, f) != 1) return 2;
    const int rows = h[0], cols = h[1], size = h[2], L = h[3];
    const bool u8 = (h[19] & 1) != 0;
    const int scheme = h[19] >> 1;

    // MOSAIC_HOST_DEVICES="0,0": the single-process multi-engine path (one engine per listed device id)
    std::vector<int> devices;
    if (const char* list = getenv("MOSAIC_HOST_DEVICES"))
        for (const char* p = list; *p;) { devices.push_back((int)strtol(p, const_cast<char**>(&p), 10)); if (*p == ',') ++p; }
    if (devices.empty()) devices.push_back(0);
    HIPPhotomosaicGenerator generator(devices);
    if (const char* k = getenv("MOSAIC_HOST_SPEC_K")) generator.setEngineOption("spec_k", atoi(k));
    Image8 img8{rows, cols, {}};
    std::vector<uint8_t> mask((size_t)size * size);
    if (u8) {
        img8.data.resize((size_t)rows * cols * 3);
        if (!readAll(f, img8.data)) return 2;
        std::vector<Image8> lib((size_t)L, Image8{size, size, std::vector<uint8_t>((size_t)size * size * 3)});
        for (auto& im : lib) if (!readAll(f, im.data)) return 2;
        if (!readAll(f, mask)) return 2;
        generator.setMainImage(img8);
        generator.setLibrary(lib);
    } else {
        Image img{rows, cols, std::vector<float>((size_t)rows * cols * 3)};
        if (!readAll(f, img.data)) return 2;
        std::vector<Image> lib((size_t)L, Image{size, size, std::vector<float>((size_t)size * size * 3)});
        for (auto& im : lib) if (!readAll(f, im.data)) return 2;
        if (!readAll(f, mask)) return 2;
        std::vector<std::vector<uint8_t>> libBgr((size_t)L, std::vector<uint8_t>((size_t)size * size * 3));
        for (auto& im : libBgr) if (!readAll(f, im)) return 2;
        generator.setMainImage(img);
        generator.setLibrary(lib);
        generator.setLibraryBGR(libBgr);
    }
    fclose(f);

    CellShape shape(size, mask);
    shape.rowSpacing = h[7]; shape.colSpacing = h[8]; shape.alternateRowSpacing = h[9]; shape.alternateColSpacing = h[10];
    shape.alternateRowOffset = h[11]; shape.alternateColOffset = h[12];
    shape.alternateColFlipHorizontal = h[13]; shape.alternateColFlipVertical = h[14];
    shape.alternateRowFlipHorizontal = h[15]; shape.alternateRowFlipVertical = h[16];
    CellGroup group;
    group.setCellShape(shape);
    group.setSizeSteps((size_t)h[18]);
    group.setDetail(h[17]);

    generator.setColourDifference(static_cast<ColourDifference::Type>(h[4]));
    generator.setColourScheme(static_cast<ColourScheme::Type>(scheme));
    generator.setCellGroup(group);
    generator.setGridState(GridUtility::getGridState(group, u8 ? &img8 : nullptr, rows, cols));
    generator.setRepeat(h[5], h[6]);
    int last = -1, monotone = 1;
    generator.progress = [&](int v) { if (v < last) monotone = 0; last = v; };
    if (!generator.generateBestFits()) { fprintf(stderr, "generateBestFits failed: %s\n", generator.lastError().c_str()); return 1; }
    if (!monotone || last != generator.getMaxProgress()) { fprintf(stderr, "progress not monotone / incomplete (%d of %d)\n", last, generator.getMaxProgress()); return 1; }

    const GridUtility::MosaicBestFit fits = generator.getBestFits();
    FILE* o = fopen(argv[2], "wb");
    const int32_t steps = (int32_t)fits.size();
    fwrite(&steps, sizeof steps, 1, o);
    for (const auto& grid : fits) {
        const int32_t dims[2] = {(int32_t)grid.size(), (int32_t)grid.at(0).size()};
        fwrite(dims, sizeof dims, 1, o);
        for (const auto& row : grid)
            for (const auto& cell : row) {
                const int64_t v = cell.has_value() ? (int64_t)cell.value() : -1;
                fwrite(&v, sizeof v, 1, o);
            }
    }
    const uint8_t background[4] = {7, 8, 9, 10};
    const std::vector<uint8_t> mosaic = generator.buildPhotomosaic(background);
    if (mosaic.empty()) { fprintf(stderr, "buildPhotomosaic failed: %s\n", generator.lastError().c_str()); return 1; }
    fwrite(mosaic.data(), 1, mosaic.size(), o);
    fclose(o);
    return 0;
}
