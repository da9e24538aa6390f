"""Image libraries: the ``.mil`` container the reference feeds to ``setLibrary``.

Host-side mirror of the part of ``ImageLibrary`` (src/ImageLibrary/ImageLibrary.{h,cpp})
that sits directly in front of the best-fit path: the list of square BGR images at one
size and the ``.mil`` (de)serialiser, without Qt or OpenCV.  Editors,
squaring / cropping tools and the CUDA resize are out of scope.

``.mil`` (ImageLibrary.cpp:117-236): big-endian QDataStream -- u32 magic 0xADBE2480, u32
version (4..6), u32 image size, u32 count, then per image the pixels (v6: a PNG in a
QByteArray, CustomQDataStream.h:39-45; v4/v5: raw type/rows/cols + bytes, :27-37) followed
by its name as a QString.  OpenCV writes BGR mats as RGB PNGs, so decoded PNG channels are
reversed back to BGR here.
"""
import io
import struct

import numpy as np

MIL_MAGIC = 0xADBE2480       # ImageLibrary.h:13
MIL_VERSION = 6              # ImageLibrary.h:12
MIL_VERSION_ENCODED = 6      # ImageLibrary.h:18


def _put_qstring(out, text):
    raw = text.encode("utf-16-be")
    out.write(struct.pack(">I", len(raw)))
    out.write(raw)


class ImageLibrary:
    """Square u8 BGR images of one size, in library order, with their names."""

    def __init__(self, imageSize=0):
        self.m_imageSize = int(imageSize)
        self.m_names = []
        self.m_images = []

    def addImage(self, image, name=""):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        if image.shape != (self.m_imageSize, self.m_imageSize, 3):
            raise ValueError("library images must be %dx%dx3 u8 BGR" % (self.m_imageSize, self.m_imageSize))
        self.m_images.append(image)
        self.m_names.append(name)

    def getImages(self):
        """ImageLibrary::getImages (ImageLibrary.h:37): what setLibrary() receives."""
        if not self.m_images:
            return np.zeros((0, self.m_imageSize, self.m_imageSize, 3), np.uint8)
        return np.stack(self.m_images)

    def getNames(self):
        return list(self.m_names)

    def getImageSize(self):
        return self.m_imageSize

    def getNumberOfLibraryImages(self):
        return len(self.m_images)

    # -- .mil -----------------------------------------------------------------
    def saveToFile(self, filename):
        """ImageLibrary::saveToFile (ImageLibrary.cpp:117-151), version 6 (PNG-encoded)."""
        from PIL import Image
        with open(filename, "wb") as out:
            out.write(struct.pack(">IIII", MIL_MAGIC, MIL_VERSION, self.m_imageSize, len(self.m_images)))
            for image, name in zip(self.m_images, self.m_names):
                png = io.BytesIO()
                Image.fromarray(image[..., ::-1]).save(png, format="PNG")       # BGR -> RGB, as cv::imencode
                blob = png.getvalue()
                out.write(struct.pack(">I", len(blob)))
                out.write(blob)
                _put_qstring(out, name)

    @classmethod
    def loadFromFile(cls, filename):
        """ImageLibrary::loadFromFile (ImageLibrary.cpp:154-231).  Versions before 5 were
        shuffled on load by the reference (random insert order, :207-217); that shuffle is
        not reproduced -- images come back in file order."""
        data = open(filename, "rb").read()
        pos = 0

        def take(fmt):
            nonlocal pos
            v = struct.unpack_from(">" + fmt, data, pos)
            pos += struct.calcsize(">" + fmt)
            return v if len(v) > 1 else v[0]

        if len(data) < 16 or take("I") != MIL_MAGIC:
            raise ValueError("File is not a valid .mil: %s" % filename)
        version = take("I")
        if version > MIL_VERSION:
            raise ValueError(".mil uses a newer file version: %s" % filename)
        if version < 4:
            raise ValueError(".mil uses an outdated file version: %s" % filename)
        lib = cls(take("I"))
        count = take("I")
        for _ in range(count):
            if version >= MIL_VERSION_ENCODED:
                n = take("I")
                from PIL import Image
                img = np.array(Image.open(io.BytesIO(data[pos:pos + n])).convert("RGB"))[..., ::-1]
                pos += n
            else:
                _type, rows, cols = take("III")
                n = take("I")
                img = np.frombuffer(data[pos:pos + n], dtype=np.uint8).reshape(rows, cols, -1)[..., :3]
                pos += n
            n = take("I")
            name = ""
            if n != 0xFFFFFFFF:
                name = data[pos:pos + n].decode("utf-16-be")
                pos += n
            lib.addImage(img, name)
        return lib
