// HIPPhotomosaicGenerator.cpp -- see the header.  Plain C++17, links libmosaic_hip.so.
#include "HIPPhotomosaicGenerator.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <stdexcept>

#include <zlib.h>

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

CellShape CellShape::resized(int size) const
{
    if (m_size == 0 || size == m_size) return *this;
    CellShape out(size, ImageUtility::resizeArea(m_masks[0], m_size, m_size, 1, size, size));
    const double ratio = (double)size / m_size;
    out.name = name;
    out.rowSpacing = std::max((int)std::floor(rowSpacing * ratio), 1);
    out.colSpacing = std::max((int)std::floor(colSpacing * ratio), 1);
    out.alternateRowSpacing = std::max((int)std::floor(alternateRowSpacing * ratio), 1);
    out.alternateColSpacing = std::max((int)std::floor(alternateColSpacing * ratio), 1);
    out.alternateRowOffset = (int)std::floor(alternateRowOffset * ratio);
    out.alternateColOffset = (int)std::floor(alternateColOffset * ratio);
    out.alternateColFlipHorizontal = alternateColFlipHorizontal;
    out.alternateColFlipVertical = alternateColFlipVertical;
    out.alternateRowFlipHorizontal = alternateRowFlipHorizontal;
    out.alternateRowFlipVertical = alternateRowFlipVertical;
    return out;
}

// ---------------------------------------------------------------------------- CellGroup
// ---------------------------------------------------------------------------- .mcs reader
namespace {

struct BigEndianReader {
    const std::vector<uint8_t>& d;
    size_t pos = 0;
    uint32_t u32()
    {
        if (pos + 4 > d.size()) throw std::runtime_error("unexpected end of .mcs file");
        const uint32_t v = ((uint32_t)d[pos] << 24) | ((uint32_t)d[pos + 1] << 16) | ((uint32_t)d[pos + 2] << 8) | d[pos + 3];
        pos += 4;
        return v;
    }
    uint8_t u8()
    {
        if (pos + 1 > d.size()) throw std::runtime_error("unexpected end of .mcs file");
        return d[pos++];
    }
    std::vector<uint8_t> bytes(size_t n)
    {
        if (pos + n > d.size()) throw std::runtime_error("unexpected end of .mcs file");
        std::vector<uint8_t> out(d.begin() + (long)pos, d.begin() + (long)(pos + n));
        pos += n;
        return out;
    }
};

// 8-bit non-interlaced PNG (grey, grey+alpha, RGB or RGBA) -> its samples, `channels` per pixel, row-major.  zlib does the
// inflating.  `what` names the thing being read in error texts; `max_side` bounds the header's dimensions.
std::vector<uint8_t> decodePng(const std::vector<uint8_t>& png, int& rows, int& cols, int& channels, const char* what, int max_side)
{
    const std::string w(what);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (png.size() < 8 || !std::equal(sig, sig + 8, png.begin())) throw std::runtime_error(w + " is not a PNG");
    BigEndianReader r{png, 8};
    std::vector<uint8_t> idat;
    channels = 0;
    rows = cols = 0;
    for (;;) {
        const uint32_t len = r.u32();
        const std::vector<uint8_t> type = r.bytes(4);
        const std::vector<uint8_t> body = r.bytes(len);
        r.u32();                                         // CRC (zlib's own checksum guards the pixel data)
        const std::string t(type.begin(), type.end());
        if (t == "IHDR") {
            BigEndianReader h{body};
            cols = (int)h.u32(); rows = (int)h.u32();
            const int depth = h.u8(), colour = h.u8();
            h.u8(); h.u8();
            const int interlace = h.u8();
            channels = colour == 0 ? 1 : colour == 4 ? 2 : colour == 2 ? 3 : colour == 6 ? 4 : 0;
            if (depth != 8 || channels == 0 || interlace != 0) throw std::runtime_error("unsupported PNG " + w + " (need 8-bit, non-interlaced, no palette)");
        } else if (t == "IDAT") idat.insert(idat.end(), body.begin(), body.end());
        else if (t == "IEND") break;
    }
    if (channels == 0 || rows <= 0 || cols <= 0) throw std::runtime_error("PNG " + w + " without a header");
    // a header asking for more than max_side is a damaged or crafted file, not a reason to try a multi-gigabyte
    // allocation; deflate cannot shrink by more than ~1032
    if (rows > max_side || cols > max_side) throw std::runtime_error("PNG " + w + " larger than " + std::to_string(max_side) + " pixels a side");
    if (idat.empty() || ((size_t)cols * channels + 1) * (size_t)rows / 1032 > idat.size() + 64)
        throw std::runtime_error("PNG " + w + " data too short for its header");
    const size_t stride = (size_t)cols * channels;
    std::vector<uint8_t> raw((stride + 1) * (size_t)rows);
    uLongf got = (uLongf)raw.size();
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size())
        throw std::runtime_error("PNG " + w + " does not inflate");
    std::vector<uint8_t> prev(stride, 0), out((size_t)rows * stride);
    for (int y = 0; y < rows; ++y) {
        const uint8_t* line = raw.data() + (size_t)y * (stride + 1);
        uint8_t* cur = out.data() + (size_t)y * stride;
        const uint8_t* up = y > 0 ? cur - stride : prev.data();
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= (size_t)channels ? cur[x - channels] : 0, b = up[x], c = x >= (size_t)channels ? up[x - channels] : 0;
            int pred = 0;
            switch (line[0]) {                           // PNG filter types 0..4
            case 0: break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) / 2; break;
            case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                      pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
            default: throw std::runtime_error("bad PNG filter");
            }
            cur[x] = (uint8_t)(line[1 + x] + pred);
        }
    }
    return out;
}

// ... its first channel: what cv::imdecode(..., IMREAD_GRAYSCALE) of the masks QImage / OpenCV write gives for a grey image.
// (a cell mask is a few hundred pixels a side; the shipped ones: 512)
std::vector<uint8_t> decodePngFirstChannel(const std::vector<uint8_t>& png, int& rows, int& cols)
{
    int channels = 0;
    const std::vector<uint8_t> all = decodePng(png, rows, cols, channels, "mask", 16384);
    if (channels == 1) return all;
    std::vector<uint8_t> out((size_t)rows * cols);
    for (size_t k = 0; k < out.size(); ++k) out[k] = all[k * (size_t)channels];
    return out;
}

// QString of a QDataStream: u32 byte length (0xFFFFFFFF = null), UTF-16BE -> UTF-8 of the BMP code points.
std::string readQString(BigEndianReader& r)
{
    std::string text;
    const uint32_t n = r.u32();
    if (n == 0xFFFFFFFFu) return text;
    const std::vector<uint8_t> utf16 = r.bytes(n);
    for (size_t i = 0; i + 1 < utf16.size(); i += 2) {
        const unsigned cp = ((unsigned)utf16[i] << 8) | utf16[i + 1];
        if (cp < 0x80) text += (char)cp;
        else if (cp < 0x800) { text += (char)(0xC0 | (cp >> 6)); text += (char)(0x80 | (cp & 0x3F)); }
        else { text += (char)(0xE0 | (cp >> 12)); text += (char)(0x80 | ((cp >> 6) & 0x3F)); text += (char)(0x80 | (cp & 0x3F)); }
    }
    return text;
}

std::vector<uint8_t> readWholeFile(const std::string& path)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> data;
    uint8_t buf[65536];
    for (size_t n; (n = fread(buf, 1, sizeof buf, f)) > 0;) data.insert(data.end(), buf, buf + n);
    fclose(f);
    return data;
}

}  // namespace

CellShape CellShape::loadFromFile(const std::string& path)
{
    const std::vector<uint8_t> data = readWholeFile(path);
    BigEndianReader r{data};
    if (r.u32() != 0x87AECFB1u) throw std::runtime_error("File is not a valid .mcs: " + path);     // CellShape.cpp:392-395
    const uint32_t version = r.u32();
    if (version > 8) throw std::runtime_error(".mcs uses a newer file version: " + path);
    if (version < 7) throw std::runtime_error(".mcs uses an outdated file version: " + path);
    const std::string name = readQString(r);             // (shape names are short labels)
    int rows = 0, cols = 0;
    std::vector<uint8_t> mask;
    if (version >= 8) mask = decodePngFirstChannel(r.bytes(r.u32()), rows, cols);
    else {
        r.u32();                                         // cv::Mat type
        rows = (int)r.u32(); cols = (int)r.u32();
        if (rows <= 0 || cols <= 0 || rows > 16384 || cols > 16384) throw std::runtime_error("raw .mcs mask larger than 16384 pixels a side");
        mask = r.bytes(r.u32());
        if (mask.size() != (size_t)rows * cols) throw std::runtime_error("raw .mcs mask is not 8-bit single channel");
    }
    if (rows != cols) throw std::runtime_error(".mcs mask is not square");
    CellShape shape;
    shape.m_size = rows;
    shape.name = name;
    for (int m = 0; m < 4; ++m) {                        // the four pre-flipped variants, as stored (no re-threshold)
        shape.m_masks[m].resize(mask.size());
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols; ++x)
                shape.m_masks[m][(size_t)y * cols + x] = mask[(size_t)((m & 2) ? rows - 1 - y : y) * cols + ((m & 1) ? cols - 1 - x : x)];
    }
    auto i32 = [&] { return (int)r.u32(); };
    shape.rowSpacing = i32(); shape.colSpacing = i32();
    shape.alternateRowSpacing = i32(); shape.alternateColSpacing = i32();
    shape.alternateRowOffset = i32(); shape.alternateColOffset = i32();
    shape.alternateColFlipHorizontal = r.u8() != 0; shape.alternateColFlipVertical = r.u8() != 0;
    shape.alternateRowFlipHorizontal = r.u8() != 0; shape.alternateRowFlipVertical = r.u8() != 0;
    return shape;
}

// ImageLibrary::loadFromFile (ImageLibrary/ImageLibrary.cpp:154-231) without Qt / OpenCV.
ImageLibraryFile ImageLibraryFile::loadFromFile(const std::string& path)
{
    const std::vector<uint8_t> data = readWholeFile(path);
    BigEndianReader r{data};
    if (r.u32() != 0xADBE2480u) throw std::runtime_error("File is not a valid .mil: " + path);       // ImageLibrary.h:13
    const uint32_t version = r.u32();
    if (version > 6) throw std::runtime_error(".mil uses a newer file version: " + path);
    if (version < 4) throw std::runtime_error(".mil uses an outdated file version: " + path);
    ImageLibraryFile lib;
    const uint32_t size = r.u32(), count = r.u32();
    if (size == 0 || size > 16384) throw std::runtime_error(".mil image size outside 1..16384: " + path);
    // every image takes at least a length word and a name word: a count the file cannot hold is a damaged header
    if ((size_t)count * 8 > data.size()) throw std::runtime_error(".mil image count exceeds the file: " + path);
    lib.imageSize = (int)size;
    const size_t per_image = (size_t)size * size * 3;
    lib.names.reserve(count);
    for (uint32_t i = 0; i < count; ++i) {
        int rows = 0, cols = 0, channels = 0;
        std::vector<uint8_t> px;
        bool rgb_order = false;
        if (version >= 6) {                               // a PNG in a QByteArray (CustomQDataStream.h:39-45,75-81); OpenCV writes BGR mats as RGB PNGs
            px = decodePng(r.bytes(r.u32()), rows, cols, channels, "library image", 16384);
            rgb_order = true;
        } else {                                          // raw cv::Mat: type, rows, cols, QByteArray (:27-37,63-72)
            r.u32();
            rows = (int)r.u32(); cols = (int)r.u32();
            if (rows <= 0 || cols <= 0 || rows > 16384 || cols > 16384) throw std::runtime_error("raw .mil image larger than 16384 pixels a side");
            px = r.bytes(r.u32());
            if (px.size() % ((size_t)rows * cols) != 0) throw std::runtime_error("raw .mil image is not 8-bit");
            channels = (int)(px.size() / ((size_t)rows * cols));
        }
        if (rows != (int)size || cols != (int)size) throw std::runtime_error(".mil image " + std::to_string(i) + " is not " + std::to_string(size) + " pixels a side");
        if (channels != 1 && channels != 3 && channels != 4) throw std::runtime_error(".mil image " + std::to_string(i) + " has " + std::to_string(channels) + " channels");
        const size_t at = lib.bgr.size();
        lib.bgr.resize(at + per_image);
        uint8_t* out = lib.bgr.data() + at;
        for (size_t k = 0; k < (size_t)rows * cols; ++k) {
            const uint8_t* p = px.data() + k * (size_t)channels;
            if (channels == 1) { out[3 * k] = out[3 * k + 1] = out[3 * k + 2] = p[0]; }
            else if (rgb_order) { out[3 * k] = p[2]; out[3 * k + 1] = p[1]; out[3 * k + 2] = p[0]; }
            else { out[3 * k] = p[0]; out[3 * k + 1] = p[1]; out[3 * k + 2] = p[2]; }
        }
        lib.names.push_back(readQString(r));
    }
    return lib;
}

void CellGroup::setCellShape(const CellShape& shape)
{
    m_cells.at(0) = shape;
    if (!shape.empty()) setDetail(0, true);
    else { m_cells.resize(1); m_detailCells.assign(1, shape); }
}

void CellGroup::setDetail(int detailPercent, bool reset)
{
    const double newDetail = reset ? m_detail : detailPercent / 100.0;
    if (newDetail != m_detail || reset) {
        m_detail = newDetail;
        if (!m_cells.at(0).empty())
            m_detailCells.at(0) = m_cells.at(0).resized(std::max((int)(m_cells.at(0).getSize() * m_detail), 1));
        setSizeSteps(m_sizeSteps, true);
    }
}

void CellGroup::setSizeSteps(size_t steps, bool reset)
{
    m_cells.resize(steps + 1);
    m_detailCells.resize(steps + 1);
    int cellSize = m_cells.at(0).getSize();
    for (size_t step = 1; step <= steps; ++step) {
        cellSize /= 2;
        if (reset || step > m_sizeSteps) {
            m_cells.at(step) = m_cells.at(step - 1).resized(cellSize);
            m_detailCells.at(step) = m_cells.at(step).resized(std::max((int)(cellSize * m_detail), 1));
        }
    }
    m_sizeSteps = steps;
}

// ---------------------------------------------------------------------------- ImageUtility
namespace ImageUtility {

namespace {
struct AreaEntry { int di, si; float alpha; };

// computeResizeAreaTab (OpenCV imgproc resize.cpp)
std::vector<AreaEntry> areaTable(int ssize, int dsize, double scale)
{
    std::vector<AreaEntry> tab;
    for (int dx = 0; dx < dsize; ++dx) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) tab.push_back({dx, sx1 - 1, (float)((sx1 - fsx1) / cell)});
        for (int sx = sx1; sx < sx2; ++sx) tab.push_back({dx, sx, (float)(1.0 / cell)});
        if (fsx2 - sx2 > 1e-3) tab.push_back({dx, sx2, (float)(std::min(std::min(fsx2 - sx2, 1.0), cell) / cell)});
    }
    return tab;
}

uint8_t roundToU8(float v) { return (uint8_t)std::clamp((int)std::lrintf(v), 0, 255); }
}  // namespace

std::vector<uint8_t> resizeArea(const std::vector<uint8_t>& src, int rows, int cols, int cn, int outRows, int outCols)
{
    if (rows == outRows && cols == outCols) return src;
    if (outRows > rows || outCols > cols) throw std::invalid_argument("INTER_CUBIC enlargement is outside the hot path");
    std::vector<uint8_t> dst((size_t)outRows * outCols * cn);
    if (rows % outRows == 0 && cols % outCols == 0 && rows / outRows == cols / outCols) {
        const int f = rows / outRows;                       // whole factor: ResizeAreaFast
        const float scale = 1.0f / (float)(f * f);
        for (int r = 0; r < outRows; ++r)
            for (int c = 0; c < outCols; ++c)
                for (int ch = 0; ch < cn; ++ch) {
                    int sum = 0;
                    for (int i = 0; i < f; ++i)
                        for (int j = 0; j < f; ++j) sum += src[((size_t)(r * f + i) * cols + (c * f + j)) * cn + ch];
                    dst[((size_t)r * outCols + c) * cn + ch] = f == 2 ? (uint8_t)((sum + 2) >> 2) : roundToU8((float)sum * scale);
                }
        return dst;
    }
    const auto xt = areaTable(cols, outCols, 1.0 / ((double)outCols / cols));
    const auto yt = areaTable(rows, outRows, 1.0 / ((double)outRows / rows));
    std::vector<float> buf((size_t)outCols * cn), sum((size_t)outCols * cn, 0.f);
    int prev = yt.front().di;
    for (const AreaEntry& y : yt) {
        std::fill(buf.begin(), buf.end(), 0.f);
        const uint8_t* row = src.data() + (size_t)y.si * cols * cn;
        for (const AreaEntry& x : xt)
            for (int ch = 0; ch < cn; ++ch) buf[(size_t)x.di * cn + ch] += (float)row[(size_t)x.si * cn + ch] * x.alpha;
        if (y.di != prev) {
            for (size_t i = 0; i < sum.size(); ++i) { dst[(size_t)prev * outCols * cn + i] = roundToU8(sum[i]); sum[i] = y.alpha * buf[i]; }
            prev = y.di;
        } else {
            for (size_t i = 0; i < sum.size(); ++i) sum[i] += y.alpha * buf[i];
        }
    }
    for (size_t i = 0; i < sum.size(); ++i) dst[(size_t)prev * outCols * cn + i] = roundToU8(sum[i]);
    return dst;
}

double calculateEntropy(const uint8_t* bgr, int rows, int cols, const uint8_t* mask, int maskStride)
{
    if (rows <= 0 || cols <= 0) return 0;
    size_t histogram[256] = {0}, pixelCount = 0;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c)
            if (!mask || mask[(size_t)r * maskStride + c] != 0) {
                const uint8_t* p = bgr + ((size_t)r * cols + c) * 3;
                ++histogram[(p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14];    // BGR2GRAY
                ++pixelCount;
            }
    double entropy = 0;
    for (size_t value : histogram) {
        const double probability = value / (double)pixelCount;
        if (probability > 0) entropy -= probability * std::log2(probability);
    }
    return entropy;
}

}  // namespace ImageUtility

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

// The active bounds of a size step as what findCellState needs of them (GridGenerator.cpp:124-140): "does this cell rect
// share positive area with ANY bound?"  That depends on the UNION of the bounds alone, and the reference's
// GridBounds::mergeBounds (Grid/GridBounds.cpp:36-107) only ever replaces two rects by their union when that union is
// itself the same point set -- so the list is never merged here at all: the bounds are rasterised once on the grid their
// own edges span (coordinate compression, coverage marks, a summed-area table) and a query is four look-ups.
// (grid.py _meets_any_bound is the same construction; tests/test_grid_steps.py holds it to the merged list.)
BoundsUnion::BoundsUnion(const std::vector<Rect>& bounds)
{
    for (const Rect& r : bounds) {
        if (r.width <= 0 || r.height <= 0) continue;
        m_xs.push_back(r.x); m_xs.push_back(r.x + r.width);
        m_ys.push_back(r.y); m_ys.push_back(r.y + r.height);
    }
    auto distinct = [](std::vector<int>& v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); };
    distinct(m_xs);
    distinct(m_ys);
    if (m_xs.size() < 2 || m_ys.size() < 2) { m_xs.clear(); m_ys.clear(); return; }
    const size_t nx = m_xs.size() - 1, ny = m_ys.size() - 1;
    std::vector<uint8_t> covered(nx * ny, 0);
    auto slot = [](const std::vector<int>& v, int c) { return (size_t)(std::lower_bound(v.begin(), v.end(), c) - v.begin()); };
    for (const Rect& r : bounds) {
        if (r.width <= 0 || r.height <= 0) continue;
        const size_t i0 = slot(m_xs, r.x), i1 = slot(m_xs, r.x + r.width), j0 = slot(m_ys, r.y), j1 = slot(m_ys, r.y + r.height);
        for (size_t j = j0; j < j1; ++j) std::fill(covered.begin() + (long)(j * nx + i0), covered.begin() + (long)(j * nx + i1), 1);
    }
    m_area.assign((nx + 1) * (ny + 1), 0);             // covered elementary boxes above and to the left of (j, i)
    for (size_t j = 0; j < ny; ++j)
        for (size_t i = 0; i < nx; ++i)
            m_area[(j + 1) * (nx + 1) + i + 1] = covered[j * nx + i] + m_area[j * (nx + 1) + i + 1] +
                                                 m_area[(j + 1) * (nx + 1) + i] - m_area[j * (nx + 1) + i];
}

bool BoundsUnion::meets(const Rect& q) const
{
    if (m_xs.empty() || q.width <= 0 || q.height <= 0) return false;
    const size_t nx = m_xs.size() - 1, ny = m_ys.size() - 1;
    // elementary boxes [first, last) that overlap the open interval (lo, hi) with positive length
    auto span = [](const std::vector<int>& v, int lo, int hi, size_t n, size_t& first, size_t& last) {
        const long a = (std::upper_bound(v.begin(), v.end(), lo) - v.begin()) - 1;       // last edge <= lo
        const long b = std::lower_bound(v.begin(), v.end(), hi) - v.begin();             // first edge >= hi
        first = (size_t)std::clamp<long>(a, 0, (long)n);
        last = std::max(first, (size_t)std::clamp<long>(b, 0, (long)n));
    };
    size_t i0, i1, j0, j1;
    span(m_xs, q.x, q.x + q.width, nx, i0, i1);
    span(m_ys, q.y, q.y + q.height, ny, j0, j1);
    const size_t w = nx + 1;
    return m_area[j1 * w + i1] - m_area[j0 * w + i1] - m_area[j1 * w + i0] + m_area[j0 * w + i0] > 0;
}

// The local bound at the detail size (PhotomosaicGeneratorBase.cpp:318-326 = GridGenerator.cpp:156-170).
Rect detailBound(const Rect& local, double detail, int size)
{
    Rect b{ std::min(size - 1, (int)(local.x * detail)), std::min(size - 1, (int)(local.y * detail)),
            std::max(1, (int)(local.width * detail)), std::max(1, (int)(local.height * detail)) };
    b.width = std::min(size - b.x, b.width);
    b.height = std::min(size - b.y, b.height);
    return b;
}

// GridGenerator::findCellState (GridGenerator.cpp:113-193)
static std::pair<CellBestFit, bool> findCellState(const CellGroup& cells, const Image8* mainImage, int x, int y,
                                                  const BoundsUnion& bounds, size_t step)
{
    const Rect g = getRectAt(cells.getCell(step), x, y);
    if (!bounds.meets(g)) return {std::nullopt, false};             // GridGenerator.cpp:124-140
    if (mainImage && step < cells.getSizeSteps()) {
        const int yStart = std::clamp(g.y, 0, mainImage->rows), yEnd = std::clamp(g.y + g.height, 0, mainImage->rows);
        const int xStart = std::clamp(g.x, 0, mainImage->cols), xEnd = std::clamp(g.x + g.width, 0, mainImage->cols);
        const int size = cells.getCellSize(step, true);
        const Rect b = detailBound(Rect{xStart - g.x, yStart - g.y, xEnd - xStart, yEnd - yStart}, cells.getDetail(), size);
        const FlipState flip = getFlipStateAt(cells.getCell(step), x, y);
        std::vector<uint8_t> cell((size_t)(yEnd - yStart) * (xEnd - xStart) * 3);
        for (int r = yStart; r < yEnd; ++r)
            std::copy_n(mainImage->data.data() + ((size_t)r * mainImage->cols + xStart) * 3, (size_t)(xEnd - xStart) * 3,
                        cell.data() + (size_t)(r - yStart) * (xEnd - xStart) * 3);
        if (yEnd - yStart != b.height || xEnd - xStart != b.width)          // resizeImage(EXACT), :180-181
            cell = ImageUtility::resizeArea(cell, yEnd - yStart, xEnd - xStart, 3, b.height, b.width);
        const uint8_t* mask = cells.getCell(step, true).getCellMask(flip.horizontal, flip.vertical).data() + (size_t)b.y * size + b.x;
        if (ImageUtility::calculateEntropy(cell.data(), b.height, b.width, mask, size) >= ImageUtility::MAX_ENTROPY * 0.7)
            return {std::nullopt, true};
    }
    return {0, false};
}

MosaicBestFit getGridState(const CellGroup& cells, const Image8* mainImage, int imageRows, int imageCols)
{
    MosaicBestFit gridState;
    if (cells.getCell(0).empty()) return gridState;
    const int gridHeight = mainImage ? mainImage->rows : imageRows, gridWidth = mainImage ? mainImage->cols : imageCols;
    std::vector<Rect> bounds[2];
    int active = 0;
    bounds[active].push_back(Rect{0, 0, gridWidth, gridHeight});
    for (size_t step = 0; step <= cells.getSizeSteps() && !bounds[active].empty(); ++step) {
        const Point g = calculateGridSize(cells.getCell(step), gridWidth, gridHeight, PAD_GRID);
        gridState.emplace_back((size_t)g.y, std::vector<CellBestFit>((size_t)g.x));
        bounds[!active].clear();
        const BoundsUnion inside(bounds[active]);           // (unmerged: the in-bounds test reads their union only)
        for (int y = -PAD_GRID; y < g.y - PAD_GRID; ++y)
            for (int x = -PAD_GRID; x < g.x - PAD_GRID; ++x) {
                const auto [bestFit, entropy] = findCellState(cells, mainImage, x, y, inside, step);
                gridState[step][(size_t)(y + PAD_GRID)][(size_t)(x + PAD_GRID)] = bestFit;
                if (!entropy) continue;
                const Rect r = getRectAt(cells.getCell(step), x, y);
                const int y0 = std::clamp(r.y, 0, gridHeight), y1 = std::clamp(r.y + r.height, 0, gridHeight);
                const int x0 = std::clamp(r.x, 0, gridWidth), x1 = std::clamp(r.x + r.width, 0, gridWidth);
                if (y0 == y1 || x0 == x1) continue;
                bounds[!active].push_back(Rect{x0, y0, x1 - x0, y1 - y0});
            }
        active = !active;
    }
    return gridState;
}

}  // namespace GridUtility

// ---------------------------------------------------------------------------- generator
HIPPhotomosaicGenerator::HIPPhotomosaicGenerator(int device) : HIPPhotomosaicGenerator(std::vector<int>{device}) {}

HIPPhotomosaicGenerator::HIPPhotomosaicGenerator(const std::vector<int>& devices)
{
    if (devices.empty()) throw std::invalid_argument("HIP engine: no device given");
    if (devices.size() == 1) {
        mm_engine* e = nullptr;
        if (mm_create(devices[0], &e) != MM_OK) throw std::runtime_error(std::string("HIP engine: ") + mm_last_error(nullptr));
        m_engines.push_back(e);
    } else {
        // several devices: the library's group (one engine per device, RCCL all-gather of the candidate lists)
        if (mm_group_create(devices.data(), (int)devices.size(), &m_group) != MM_OK)
            throw std::runtime_error(std::string("HIP engine group: ") + mm_group_last_error(nullptr));
        for (int i = 0; i < mm_group_size(m_group); ++i) m_engines.push_back(mm_group_engine(m_group, i));
    }
    m_engine = m_engines.front();
}

HIPPhotomosaicGenerator::~HIPPhotomosaicGenerator()
{
    if (m_group) mm_group_destroy(m_group);                 // (owns its engines)
    else for (mm_engine* e : m_engines) mm_destroy(e);
}

bool HIPPhotomosaicGenerator::setEngineOption(const char* name, int value)
{
    if (m_group) {
        if (mm_group_set_option(m_group, name, value) != MM_OK) { m_lastError = mm_group_last_error(m_group); return false; }
        return true;
    }
    for (mm_engine* e : m_engines)
        if (mm_set_option(e, name, value) != MM_OK) { m_lastError = mm_last_error(e); return false; }
    return true;
}

void HIPPhotomosaicGenerator::setMainImage(const Image& img) { m_img = img; m_img8 = Image8{}; m_rows = img.rows; m_cols = img.cols; m_staleInputs = true; }
void HIPPhotomosaicGenerator::setLibrary(const std::vector<Image>& lib) { m_lib = lib; m_lib8.clear(); m_libCount = lib.size(); m_staleInputs = true; }
void HIPPhotomosaicGenerator::setMainImage(const Image8& img) { m_img8 = img; m_img = Image{}; m_rows = img.rows; m_cols = img.cols; m_staleInputs = true; }
void HIPPhotomosaicGenerator::setLibrary(const std::vector<Image8>& lib)
{
    m_lib8 = lib; m_lib.clear(); m_libCount = lib.size(); m_staleInputs = true;
    m_libBgr.clear();
    for (const Image8& im : lib) m_libBgr.push_back(im.data);      // buildPhotomosaic paints with the same images
    m_staleBgr = true;
}
void HIPPhotomosaicGenerator::setCellGroup(const CellGroup& cellGroup) { m_cells = cellGroup; m_staleInputs = true; }
void HIPPhotomosaicGenerator::setGridState(const GridUtility::MosaicBestFit& gridState) { m_bestFits = gridState; }
void HIPPhotomosaicGenerator::setRepeat(int repeatRange, int repeatAddition) { m_repeatRange = repeatRange; m_repeatAddition = repeatAddition; }

void HIPPhotomosaicGenerator::setColourDifference(ColourDifference::Type type)
{
    if (type == ColourDifference::Type::MAX)
        throw std::invalid_argument("No function for given type");      // ColourDifference.cpp:23
    m_colourDiffType = type;
}

std::vector<float> ColourScheme::rotations(Type type)
{
    switch (type) {
    case Type::NONE: return {};
    case Type::COMPLEMENTARY: return {180.0f};                 // ColourScheme.cpp:44-60
    case Type::TRIADIC: return {120.0f, 240.0f};               // :66-87
    case Type::COMPOUND: return {150.0f, 210.0f};              // :93-114
    case Type::TETRADIC: return {90.0f, 180.0f, 270.0f};       // :121-146
    case Type::ANALAGOUS: return {30.0f, 60.0f, 90.0f};        // :153-177
    default: throw std::invalid_argument("No function for given type");      // ColourScheme.cpp:31
    }
}

void HIPPhotomosaicGenerator::setColourScheme(ColourScheme::Type type)
{
    ColourScheme::rotations(type);          // throws on MAX
    m_colourSchemeType = type;
    m_staleInputs = true;
}

std::vector<Image8> HIPPhotomosaicGenerator::mainImageVariants()
{
    std::vector<Image8> out;
    if (m_img8.data.empty()) { m_lastError = "colour-scheme variants need the u8 BGR main image"; return out; }
    out.push_back(m_img8);
    for (float degrees : ColourScheme::rotations(m_colourSchemeType)) {
        Image8 v{m_img8.rows, m_img8.cols, std::vector<uint8_t>(m_img8.data.size())};
        if (mm_hue_rotate_u8(m_engine, m_img8.data.data(), m_img8.rows, m_img8.cols, degrees, v.data.data()) != MM_OK) {
            m_lastError = mm_last_error(m_engine);
            return {};
        }
        out.push_back(std::move(v));
    }
    return out;
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

std::vector<std::pair<std::string, double>> HIPPhotomosaicGenerator::timingLog() const
{
    const mm_timing t = timing();
    mm_stage_timing st{};
    mm_get_stage_timing(m_engine, &st);
    return {{"generateBestFits", t.total_ms}, {"Preprocess", st.preprocess_ms}, {"cudaMalloc", st.alloc_ms + st.pinned_ms},
            {"cudaMemcpy", t.stage_ms}, {"DiffReduce", t.seed_ms + t.score_ms + t.audit_ms + t.rescore_ms}, {"Flatten", t.select_ms},
            {"Repeats+FindLowest", t.resolve_ms}};
}

// One record per cell of a size step, raster order (CPUPhotomosaicGenerator.cpp:63-77).
// painting = false: cells with a value, bound at the detail size (getCellAt,
// PhotomosaicGeneratorBase.cpp:299-326); painting = true: the same cells with the bound of the
// NORMAL cell at full size (buildPhotomosaic, :152-164).
std::vector<mm_cell> HIPPhotomosaicGenerator::cellRecords(size_t step, bool painting) const
{
    const CellShape& normal = m_cells.getCell(step);
    const CellShape& detail = painting ? normal : m_cells.getCell(step, true);
    const int size = detail.getSize();
    const double d = painting ? 1.0 : m_cells.getDetail();
    const GridUtility::StepBestFit& grid = m_bestFits[step];
    const int gridH = (int)grid.size(), gridW = (int)grid.at(0).size();
    std::vector<mm_cell> cells;
    for (int y = -GridUtility::PAD_GRID; y < gridH - GridUtility::PAD_GRID; ++y)
        for (int x = -GridUtility::PAD_GRID; x < gridW - GridUtility::PAD_GRID; ++x) {
            if (!grid[(size_t)(y + GridUtility::PAD_GRID)][(size_t)(x + GridUtility::PAD_GRID)].has_value()) continue;
            const GridUtility::Rect r = GridUtility::getRectAt(normal, x, y);
            const int y0 = std::clamp(r.y, 0, m_rows), y1 = std::clamp(r.y + r.height, 0, m_rows);
            const int x0 = std::clamp(r.x, 0, m_cols), x1 = std::clamp(r.x + r.width, 0, m_cols);
            if (y0 == y1 || x0 == x1) continue;                  // rect misses the image (never has a value)
            const GridUtility::Rect b = GridUtility::detailBound(GridUtility::Rect{x0 - r.x, y0 - r.y, x1 - x0, y1 - y0}, d, size);
            const GridUtility::FlipState f = GridUtility::getFlipStateAt(detail, x, y);
            cells.push_back(mm_cell{ r.x, r.y, b.y, b.y + b.height, b.x, b.x + b.width, (f.horizontal ? 1 : 0) + (f.vertical ? 2 : 0),
                                     x + GridUtility::PAD_GRID, y + GridUtility::PAD_GRID });
        }
    return cells;
}

static std::vector<uint8_t> masks4(const CellShape& shape)
{
    std::vector<uint8_t> masks;
    for (int id = 0; id < 4; ++id) {                                  // order of CUDAPhotomosaicGenerator.cpp:122-134
        const auto& m = shape.getCellMask(id & 1, id & 2);
        masks.insert(masks.end(), m.begin(), m.end());
    }
    return masks;
}

static std::vector<int64_t> flatGrid(const GridUtility::StepBestFit& grid)
{
    const size_t gridH = grid.size(), gridW = grid.at(0).size();
    std::vector<int64_t> fits(gridW * gridH, -1);
    for (size_t gy = 0; gy < gridH; ++gy)
        for (size_t gx = 0; gx < gridW; ++gx)
            if (grid[gy][gx].has_value()) fits[gy * gridW + gx] = (int64_t)grid[gy][gx].value();
    return fits;
}

// PhotomosaicGeneratorBase::buildPhotomosaic (:110-207): per size step the NORMAL masks, the u8
// library halved once more (:126-127), and only the pixels no earlier step covers (:189-196).
std::vector<uint8_t> HIPPhotomosaicGenerator::buildPhotomosaic(const uint8_t backgroundColour[4])
{
    m_lastError.clear();
    if (m_bestFits.empty() || m_libBgr.empty()) { m_lastError = "buildPhotomosaic needs best fits and the u8 library"; return {}; }
    const int size = m_cells.getCellSize(0);
    std::vector<uint8_t> flat;
    flat.reserve(m_libBgr.size() * (size_t)size * size * 3);
    for (const auto& im : m_libBgr) {
        if (im.size() != (size_t)size * size * 3) { m_lastError = "library image size != cell size"; return {}; }
        flat.insert(flat.end(), im.begin(), im.end());
    }
    mm_status st;
    if ((st = mm_set_library_bgr(m_engine, flat.data(), (int)m_libBgr.size(), size)) != MM_OK) { m_lastError = mm_last_error(m_engine); return {}; }
    m_staleInputs = true;               // the masks staged for generating are replaced below
    std::vector<uint8_t> mosaic((size_t)m_rows * m_cols * 4);
    for (size_t step = 0; step < m_bestFits.size(); ++step) {
        const std::vector<uint8_t> masks = masks4(m_cells.getCell(step));
        if ((step != 0 && (st = mm_shrink_library_bgr(m_engine, 2)) != MM_OK) ||
            (st = mm_set_masks(m_engine, masks.data(), m_cells.getCellSize(step))) != MM_OK) { m_lastError = mm_last_error(m_engine); return {}; }
        const GridUtility::StepBestFit& grid = m_bestFits[step];
        const std::vector<int64_t> fits = flatGrid(grid);
        const std::vector<mm_cell> cells = cellRecords(step, true);
        st = mm_build_photomosaic_step(m_engine, cells.data(), (int)cells.size(), (int)grid.at(0).size(), (int)grid.size(),
                                       fits.data(), backgroundColour, m_rows, m_cols, (int)step,
                                       step + 1 == m_bestFits.size() ? mosaic.data() : nullptr);
        if (st != MM_OK) { m_lastError = mm_last_error(m_engine); return {}; }
    }
    return mosaic;
}

namespace {
struct ProgressScope { HIPPhotomosaicGenerator* self; int base, weight; };
void onProgress(int done, void* user)
{
    auto* scope = static_cast<ProgressScope*>(user);
    if (scope->self->progress) scope->self->progress(scope->base + done * scope->weight);
}
}  // namespace

// preprocessMainImage / preprocessLibraryImages (PhotomosaicGeneratorBase.cpp:223-290) -> device.
bool HIPPhotomosaicGenerator::stageInputs()
{
    // library and target are replicated on every engine (<= ~1.1 GB even for an 8192^2 target with
    // 10 000 images: nothing against 288 GB of HBM per GPU)
    mm_engine* const first = m_engine;
    bool ok = true;
    for (size_t i = 0; i < m_engines.size() && ok; ++i) {
        m_engine = m_engines[i];
        ok = stageInputsOn();
    }
    m_engine = first;
    return ok;
}

bool HIPPhotomosaicGenerator::stageInputsOn()
{
    const int detailSize = m_cells.getCellSize(0, true);
    const int type = (int)m_colourDiffType;
    mm_status st = MM_OK;
    if (!m_lib8.empty()) {
        const int size = m_lib8.front().rows;
        if (size < detailSize) { m_lastError = "library images smaller than the detail cell (INTER_CUBIC enlargement is outside the hot path)"; return false; }
        std::vector<uint8_t> flat;
        flat.reserve(m_lib8.size() * (size_t)size * size * 3);
        for (const Image8& im : m_lib8) {
            if (im.rows != size || im.cols != size) { m_lastError = "library images differ in size"; return false; }
            flat.insert(flat.end(), im.data.begin(), im.data.end());
        }
        st = mm_set_library_u8_sized(m_engine, flat.data(), (int)m_lib8.size(), size, detailSize, type);
    } else if (!m_lib.empty()) {
        std::vector<float> flat((size_t)m_lib.size() * detailSize * detailSize * 3);
        for (size_t i = 0; i < m_lib.size(); ++i) {
            if (m_lib[i].rows != detailSize || m_lib[i].cols != detailSize) { m_lastError = "library image size != detail cell size"; return false; }
            std::copy(m_lib[i].data.begin(), m_lib[i].data.end(), flat.begin() + i * (size_t)detailSize * detailSize * 3);
        }
        st = mm_set_library(m_engine, flat.data(), (int)m_lib.size(), detailSize);
    } else { m_lastError = "no library"; return false; }
    if (st != MM_OK) { m_lastError = mm_last_error(m_engine); return false; }
    if (m_colourSchemeType != ColourScheme::Type::NONE) {
        // every variant's cell aliases the last variant (setColourScheme): compare against that one
        const std::vector<Image8> variants = mainImageVariants();
        if (variants.empty()) return false;
        st = mm_set_target_u8(m_engine, variants.back().data.data(), m_img8.rows, m_img8.cols, type);
    }
    else if (!m_img8.data.empty()) st = mm_set_target_u8(m_engine, m_img8.data.data(), m_img8.rows, m_img8.cols, type);
    else if (!m_img.data.empty()) st = mm_set_target(m_engine, m_img.data.data(), m_img.rows, m_img.cols);
    else { m_lastError = "no main image"; return false; }
    if (st != MM_OK) { m_lastError = mm_last_error(m_engine); return false; }
    return true;
}

// The step loop of generateBestFits (CPUPhotomosaicGenerator.cpp:46-100).
bool HIPPhotomosaicGenerator::generateBestFits()
{
    m_lastError.clear();
    auto fail = [&](mm_status st) {
        if (st != MM_CANCELLED) m_lastError = mm_last_error(m_engine);   // the reference shows a message box + LogCritical
        return false;
    };
    const size_t steps = m_bestFits.size();
    if (steps == 0 || steps > m_cells.getSizeSteps() + 1) { m_lastError = "grid state does not match the cell group's size steps"; return false; }
    // the resident library is halved per step, so several steps re-stage it every time
    if (m_staleInputs || steps > 1) {
        if (!stageInputs()) return false;
        m_staleInputs = steps > 1;
    }
    int progressBase = 0;
    for (size_t step = 0; step < steps; ++step) {
        if (m_wasCanceled) return false;
        const CellShape& normal = m_cells.getCell(step);
        const CellShape& detail = m_cells.getCell(step, true);
        if (detail.getSize() <= 0 || detail.getSize() > normal.getSize()) { m_lastError = "detail cell larger than the cell"; return false; }
        mm_status st;
        const std::vector<uint8_t> masks = masks4(detail);
        for (mm_engine* e : m_engines) {
            auto failOn = [&](mm_status code) { m_lastError = mm_last_error(e); (void)code; return false; };
            if (step > 0 && (st = mm_shrink_library(e, 2)) != MM_OK) return failOn(st);       // batchResizeMat, :95-99
            if ((st = mm_set_masks(e, masks.data(), detail.getSize())) != MM_OK) return failOn(st);
            if ((st = mm_set_cell_resize(e, normal.getSize(), detail.getSize())) != MM_OK) return failOn(st);   // getCellAt's resize, :313-315
        }

        GridUtility::StepBestFit& grid = m_bestFits[step];
        const int gridH = (int)grid.size(), gridW = (int)grid.at(0).size();
        const std::vector<mm_cell> cells = cellRecords(step, false);
        std::vector<int64_t> bestFit((size_t)gridW * gridH);
        ProgressScope scope{this, progressBase, (int)std::pow(4, (steps - 1) - step)};        // progressStep, :54
        if (m_engines.size() > 1) {
            if (!generateStepSharded(cells, gridW, gridH, bestFit)) return false;
            if (progress) progress(progressBase + (int)cells.size() * scope.weight);
        } else {
            st = mm_generate(m_engine, cells.data(), (int)cells.size(), gridW, gridH, (int)m_colourDiffType,
                             m_repeatRange, m_repeatAddition, bestFit.data(), nullptr,
                             progress ? onProgress : nullptr, &scope, &m_wasCanceled);
            if (st != MM_OK) return fail(st);
        }
        progressBase += gridW * gridH * scope.weight;

        // Write back where the grid state has a value (CUDAPhotomosaicGenerator.cpp:310-324);
        // the sentinel ("Failed to find a best fit", CPUPhotomosaicGenerator.cpp:173-177) removes the cell.
        for (int gy = 0; gy < gridH; ++gy)
            for (int gx = 0; gx < gridW; ++gx) {
                auto& cell = grid[(size_t)gy][(size_t)gx];
                if (!cell.has_value()) continue;
                const int64_t v = bestFit[(size_t)gy * gridW + gx];
                if (v >= 0 && v < (int64_t)m_libCount) cell = (size_t)v; else cell = std::nullopt;
            }
    }
    for (mm_engine* e : m_engines) mm_set_cell_scale(e, 1);
    if (progress) progress(getMaxProgress());
    return true;
}

// One size step over several devices: the library's group does the sharding, the ONE all-gather of the candidate
// lists (RCCL; host memory where RCCL refuses the device list) and the verifying resolve (mm_group_generate).
bool HIPPhotomosaicGenerator::generateStepSharded(const std::vector<mm_cell>& cells, int gridW, int gridH,
                                                  std::vector<int64_t>& bestFit)
{
    const mm_status st = mm_group_generate(m_group, cells.data(), (int)cells.size(), gridW, gridH, (int)m_colourDiffType,
                                           m_repeatRange, m_repeatAddition, bestFit.data(), nullptr, nullptr, nullptr, &m_wasCanceled);
    if (st != MM_OK && st != MM_CANCELLED) m_lastError = mm_group_last_error(m_group);
    return st == MM_OK;
}
