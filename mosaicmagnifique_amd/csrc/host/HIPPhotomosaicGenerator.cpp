// HIPPhotomosaicGenerator.cpp -- see the header.  Plain C++17, links libmosaic_hip.so.
#include "HIPPhotomosaicGenerator.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>

// ---------------------------------------------------------------------------- CellShape
CellShape::CellShape(int size, std::vector<uint8_t> mask) : m_size(size)
{
    if ((int)mask.size() != size * size) throw std::invalid_argument("Unsupported mask type");
    rowSpacing = colSpacing = alternateRowSpacing = alternateColSpacing = size;   // CellShape.cpp:36-44
    for (auto& v : mask) v = v > 127 ? 255 : 0;                                   // CellShape.cpp:118-123
    for (int fv = 0; fv < 2; ++fv)
        for (int fh = 0; fh < 2; ++fh) {
            auto& m = m_masks[fh + 2 * fv];
            m.resize(mask.size());
            for (int r = 0; r < size; ++r)
                for (int c = 0; c < size; ++c)
                    m[(size_t)r * size + c] = mask[(size_t)(fv ? size - 1 - r : r) * size + (fh ? size - 1 - c : c)];
        }
}

const std::vector<uint8_t>& CellShape::getCellMask(bool flippedHorizontal, bool flippedVertical) const
{
    return m_masks[(flippedHorizontal ? 1 : 0) + (flippedVertical ? 2 : 0)];
}

// ---------------------------------------------------------------------------- GridUtility
namespace GridUtility {

static int cellsAlong(int extent, int spacing, int altSpacing)      // GridUtility.cpp:25-52
{
    if (spacing != altSpacing) return 2 * ((extent + spacing + altSpacing - 1) / (spacing + altSpacing));
    return (extent + spacing - 1) / spacing;
}

Point calculateGridSize(const CellShape& s, int imageWidth, int imageHeight, int pad)
{
    return { cellsAlong(imageWidth, s.colSpacing, s.alternateColSpacing) + pad,
             cellsAlong(imageHeight, s.rowSpacing, s.alternateRowSpacing) + pad };
}

static int axisStart(int n, int spacing, int altSpacing)            // GridUtility.cpp:86-115
{
    const int half = n / 2, rest = n - half;                        // C++ truncation toward zero
    return n < 0 ? rest * spacing + half * altSpacing : half * spacing + rest * altSpacing;
}

Rect getRectAt(const CellShape& s, int x, int y)
{
    Rect r;
    r.x = axisStart(x, s.colSpacing, s.alternateColSpacing) + ((y % 2 != 0) ? s.alternateRowOffset : 0);
    r.y = axisStart(y, s.rowSpacing, s.alternateRowSpacing) + ((x % 2 != 0) ? s.alternateColOffset : 0);
    r.width = r.height = s.getSize();
    return r;
}

FlipState getFlipStateAt(const CellShape& s, int x, int y)         // GridUtility.cpp:118-132
{
    FlipState f;
    if (s.alternateColFlipHorizontal && x % 2 != 0) f.horizontal = !f.horizontal;
    if (s.alternateRowFlipHorizontal && y % 2 != 0) f.horizontal = !f.horizontal;
    if (s.alternateColFlipVertical && x % 2 != 0) f.vertical = !f.vertical;
    if (s.alternateRowFlipVertical && y % 2 != 0) f.vertical = !f.vertical;
    return f;
}

MosaicBestFit getGridState(const CellGroup& cells, int imageRows, int imageCols)
{
    const CellShape& shape = cells.getCell(0);
    MosaicBestFit state;
    if (shape.getSize() == 0) return state;
    const Point g = calculateGridSize(shape, imageCols, imageRows, PAD_GRID);
    state.emplace_back((size_t)g.y, std::vector<CellBestFit>((size_t)g.x));
    for (int y = -PAD_GRID; y < g.y - PAD_GRID; ++y)
        for (int x = -PAD_GRID; x < g.x - PAD_GRID; ++x) {
            const Rect r = getRectAt(shape, x, y);
            const int y0 = std::clamp(r.y, 0, imageRows), y1 = std::clamp(r.y + r.height, 0, imageRows);
            const int x0 = std::clamp(r.x, 0, imageCols), x1 = std::clamp(r.x + r.width, 0, imageCols);
            if (y0 != y1 && x0 != x1) state[0][(size_t)(y + PAD_GRID)][(size_t)(x + PAD_GRID)] = 0;
        }
    return state;
}

}  // namespace GridUtility

// ---------------------------------------------------------------------------- generator
HIPPhotomosaicGenerator::HIPPhotomosaicGenerator(int device)
{
    if (mm_create(device, &m_engine) != MM_OK)
        throw std::runtime_error(std::string("HIP engine: ") + mm_last_error(nullptr));
}

HIPPhotomosaicGenerator::~HIPPhotomosaicGenerator() { mm_destroy(m_engine); }

void HIPPhotomosaicGenerator::setMainImage(const Image& img) { m_img = img; m_staleInputs = true; }
void HIPPhotomosaicGenerator::setLibrary(const std::vector<Image>& lib) { m_lib = lib; m_staleInputs = true; }
void HIPPhotomosaicGenerator::setCellGroup(const CellGroup& cellGroup) { m_cells = cellGroup; m_staleInputs = true; }
void HIPPhotomosaicGenerator::setGridState(const GridUtility::MosaicBestFit& gridState) { m_bestFits = gridState; }
void HIPPhotomosaicGenerator::setRepeat(int repeatRange, int repeatAddition) { m_repeatRange = repeatRange; m_repeatAddition = repeatAddition; }

void HIPPhotomosaicGenerator::setColourDifference(ColourDifference::Type type)
{
    if (type == ColourDifference::Type::MAX)
        throw std::invalid_argument("No function for given type");      // ColourDifference.cpp:23
    m_colourDiffType = type;
}

int HIPPhotomosaicGenerator::getMaxProgress() const                     // PhotomosaicGeneratorBase.cpp:210-214
{
    return (int)(std::pow(4, m_bestFits.size() - 1) * m_bestFits.size() * m_bestFits.at(0).at(0).size() * m_bestFits.at(0).size());
}

mm_timing HIPPhotomosaicGenerator::timing() const
{
    mm_timing t{};
    mm_get_timing(m_engine, &t);
    return t;
}

// One record per valid cell, raster order (CPUPhotomosaicGenerator.cpp:63-77); the bound
// is getCellAt's at detail 1 (PhotomosaicGeneratorBase.cpp:299-326).
std::vector<mm_cell> HIPPhotomosaicGenerator::cellRecords(bool onlyWithValue) const
{
    const CellShape& normal = m_cells.getCell(0);
    const CellShape& detail = m_cells.getCell(0, true);
    const int size = detail.getSize();
    const GridUtility::StepBestFit& grid = m_bestFits[0];
    const int gridH = (int)grid.size(), gridW = (int)grid.at(0).size();
    std::vector<mm_cell> cells;
    for (int y = -GridUtility::PAD_GRID; y < gridH - GridUtility::PAD_GRID; ++y)
        for (int x = -GridUtility::PAD_GRID; x < gridW - GridUtility::PAD_GRID; ++x) {
            if (onlyWithValue && !grid[(size_t)(y + GridUtility::PAD_GRID)][(size_t)(x + GridUtility::PAD_GRID)].has_value()) continue;
            const GridUtility::Rect r = GridUtility::getRectAt(normal, x, y);
            const int y0 = std::clamp(r.y, 0, m_img.rows), y1 = std::clamp(r.y + r.height, 0, m_img.rows);
            const int x0 = std::clamp(r.x, 0, m_img.cols), x1 = std::clamp(r.x + r.width, 0, m_img.cols);
            if (y0 == y1 || x0 == x1) continue;                  // rect misses the image (never has a value)
            const double d = m_cells.getDetail();
            int bx = std::min(size - 1, (int)((x0 - r.x) * d)), by = std::min(size - 1, (int)((y0 - r.y) * d));
            int bw = std::max(1, (int)((x1 - x0) * d)), bh = std::max(1, (int)((y1 - y0) * d));
            bw = std::min(size - bx, bw);
            bh = std::min(size - by, bh);
            const GridUtility::FlipState f = GridUtility::getFlipStateAt(detail, x, y);
            cells.push_back(mm_cell{ r.x, r.y, by, by + bh, bx, bx + bw, (f.horizontal ? 1 : 0) + (f.vertical ? 2 : 0),
                                     x + GridUtility::PAD_GRID, y + GridUtility::PAD_GRID });
        }
    return cells;
}

std::vector<uint8_t> HIPPhotomosaicGenerator::buildPhotomosaic(const uint8_t backgroundColour[4])
{
    m_lastError.clear();
    if (m_bestFits.size() != 1 || m_libBgr.empty()) { m_lastError = "buildPhotomosaic needs one size step and the u8 library"; return {}; }
    const CellShape& normal = m_cells.getCell(0);
    const int size = normal.getSize();
    mm_status st;
    if (m_staleBgr) {
        std::vector<uint8_t> flat;
        flat.reserve(m_libBgr.size() * (size_t)size * size * 3);
        for (const auto& im : m_libBgr) {
            if (im.size() != (size_t)size * size * 3) { m_lastError = "library image size != cell size"; return {}; }
            flat.insert(flat.end(), im.begin(), im.end());
        }
        std::vector<uint8_t> masks;
        for (int id = 0; id < 4; ++id) { const auto& m = normal.getCellMask(id & 1, id & 2); masks.insert(masks.end(), m.begin(), m.end()); }
        if ((st = mm_set_library_bgr(m_engine, flat.data(), (int)m_libBgr.size(), size)) != MM_OK ||
            (st = mm_set_masks(m_engine, masks.data(), size)) != MM_OK) { m_lastError = mm_last_error(m_engine); return {}; }
        m_staleBgr = false;
    }
    const GridUtility::StepBestFit& grid = m_bestFits[0];
    const int gridH = (int)grid.size(), gridW = (int)grid.at(0).size();
    std::vector<int64_t> fits((size_t)gridW * gridH, -1);
    for (int gy = 0; gy < gridH; ++gy)
        for (int gx = 0; gx < gridW; ++gx)
            if (grid[(size_t)gy][(size_t)gx].has_value()) fits[(size_t)gy * gridW + gx] = (int64_t)grid[(size_t)gy][(size_t)gx].value();
    const std::vector<mm_cell> cells = cellRecords(true);
    std::vector<uint8_t> mosaic((size_t)m_img.rows * m_img.cols * 4);
    st = mm_build_photomosaic(m_engine, cells.data(), (int)cells.size(), gridW, gridH, fits.data(), backgroundColour,
                              m_img.rows, m_img.cols, mosaic.data());
    if (st != MM_OK) { m_lastError = mm_last_error(m_engine); return {}; }
    return mosaic;
}

static void onProgress(int done, void* user)
{
    auto* self = static_cast<HIPPhotomosaicGenerator*>(user);
    if (self->progress) self->progress(done);
}

bool HIPPhotomosaicGenerator::generateBestFits()
{
    m_lastError.clear();
    auto fail = [&](mm_status st) {
        if (st != MM_CANCELLED) m_lastError = mm_last_error(m_engine);   // the reference shows a message box + LogCritical
        return false;
    };
    if (m_bestFits.size() != 1) { m_lastError = "size steps > 0 are outside the hot-path scope"; return false; }
    const CellShape& normal = m_cells.getCell(0);
    const CellShape& detail = m_cells.getCell(0, true);
    const int size = detail.getSize();

    if (m_staleInputs) {        // staged once; later generateBestFits() calls reuse the resident copies
        if (m_lib.empty() || m_img.data.empty()) { m_lastError = "no main image / library"; return false; }
        std::vector<float> lib((size_t)m_lib.size() * size * size * 3);
        for (size_t i = 0; i < m_lib.size(); ++i) {
            if (m_lib[i].rows != size || m_lib[i].cols != size) { m_lastError = "library image size != cell size"; return false; }
            std::copy(m_lib[i].data.begin(), m_lib[i].data.end(), lib.begin() + i * (size_t)size * size * 3);
        }
        std::vector<uint8_t> masks;
        for (int id = 0; id < 4; ++id) {
            const auto& m = detail.getCellMask(id & 1, id & 2);           // order of CUDAPhotomosaicGenerator.cpp:122-134
            masks.insert(masks.end(), m.begin(), m.end());
        }
        mm_status st;
        if ((st = mm_set_library(m_engine, lib.data(), (int)m_lib.size(), size)) != MM_OK) return fail(st);
        if ((st = mm_set_masks(m_engine, masks.data(), size)) != MM_OK) return fail(st);
        if ((st = mm_set_target(m_engine, m_img.data.data(), m_img.rows, m_img.cols)) != MM_OK) return fail(st);
        m_staleInputs = false;
    }

    GridUtility::StepBestFit& grid = m_bestFits[0];
    const int gridH = (int)grid.size(), gridW = (int)grid.at(0).size();
    const std::vector<mm_cell> cells = cellRecords(true);

    std::vector<int64_t> bestFit((size_t)gridW * gridH);
    const mm_status st = mm_generate(m_engine, cells.data(), (int)cells.size(), gridW, gridH, (int)m_colourDiffType,
                                     m_repeatRange, m_repeatAddition, bestFit.data(), nullptr,
                                     progress ? onProgress : nullptr, this, &m_wasCanceled);
    if (st != MM_OK) return fail(st);

    // Write back where the grid state has a value (CUDAPhotomosaicGenerator.cpp:310-324);
    // the sentinel ("Failed to find a best fit", CPUPhotomosaicGenerator.cpp:173-177) removes the cell.
    for (int gy = 0; gy < gridH; ++gy)
        for (int gx = 0; gx < gridW; ++gx) {
            auto& cell = grid[(size_t)gy][(size_t)gx];
            if (!cell.has_value()) continue;
            const int64_t v = bestFit[(size_t)gy * gridW + gx];
            if (v >= 0 && v < (int64_t)m_lib.size()) cell = (size_t)v; else cell = std::nullopt;
        }
    if (progress) progress(getMaxProgress());
    return true;
}
