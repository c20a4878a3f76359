// writers.hpp — native sinks for rendered glyph files:
//   TarWriter   /root/reference/src/writer/tar.rs:30-157   (POSIX.1-1988 ustar stream)
//   FileWriter  src/writer/file.rs:10-52                    (directory on the filesystem)
// Both implement vg::Writer (font_manager.hpp; writer/mod.rs:10-19).  Errors are thrown as
// std::runtime_error (the reference returns anyhow errors; the first one aborts a render).
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "font_manager.hpp"

namespace vg {

// tar.rs:30-157.  Headers: name (100 bytes, longer names are an error, tar.rs:160-171), mode
// 0644 / 0755, uid = gid = 0, size, mtime, typeflag '0' / '5', "ustar\0" "00", checksum; file data
// padded to 512 bytes; finish() appends 1024 zero bytes.  The reference stamps every header with
// the wall clock (tar.rs:68-72), so its streams are never reproducible; here the time is a
// parameter (mtime < 0: wall clock at each header, as the reference does).
class TarWriter : public Writer {
public:
	// exactly one sink: an owned FILE* (closed in the destructor when `own`), or a memory buffer
	TarWriter(std::FILE *sink, bool own, int64_t mtime);
	TarWriter(std::vector<uint8_t> *memory, int64_t mtime);
	~TarWriter() override;
	void write_directory(const std::string &path) override;
	void write_file(const std::string &path, const std::vector<uint8_t> &data) override;
	void write_file(const std::string &path, const uint8_t *data, size_t len);
	void write_bytes(const std::string &path, const uint8_t *data, size_t len) override { write_file(path, data, len); }
	void write_gather(const std::string &path, const Piece *pieces, size_t n) override; // one header, the pieces streamed
	void finish() override;

private:
	void put(const void *p, size_t n);
	void header(const std::string &path, uint64_t size, uint64_t mode, uint8_t typeflag);
	std::FILE *file_ = nullptr;
	bool own_ = false;
	std::vector<uint8_t> *mem_ = nullptr;
	int64_t mtime_;
	bool finished_ = false;
};

// file.rs:10-52: write_file = fs::write(folder/name) (parent directories are NOT created: the
// manager writes "<font>/" first), write_directory = create_dir_all, finish = no-op.
class FileWriter : public Writer {
public:
	explicit FileWriter(std::string folder) : folder_(std::move(folder)) {}
	void write_directory(const std::string &path) override;
	void write_file(const std::string &path, const std::vector<uint8_t> &data) override;
	void write_file(const std::string &path, const uint8_t *data, size_t len);
	void write_bytes(const std::string &path, const uint8_t *data, size_t len) override { write_file(path, data, len); }
	void write_gather(const std::string &path, const Piece *pieces, size_t n) override;

private:
	std::string folder_;
};

} // namespace vg
