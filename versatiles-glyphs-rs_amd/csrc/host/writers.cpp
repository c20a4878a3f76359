// writers.cpp — see writers.hpp.
#include "writers.hpp"

#include <cerrno>
#include <cstring>
#include <ctime>
#include <stdexcept>

#include <sys/stat.h>
#include <sys/types.h>

namespace vg {

namespace {

// tar.rs:144-155: right-aligned octal digits, zero filled, the last byte of the field a space
void put_octal(uint8_t *field, size_t len, uint64_t v)
{
	field[len - 1] = ' ';
	for (size_t i = len - 1; i-- > 0;) {
		field[i] = (uint8_t)('0' + (v & 7));
		v >>= 3;
	}
}

} // namespace

TarWriter::TarWriter(std::FILE *sink, bool own, int64_t mtime) : file_(sink), own_(own), mtime_(mtime)
{
	if (!sink)
		throw std::runtime_error("tar writer: no output stream");
}

TarWriter::TarWriter(std::vector<uint8_t> *memory, int64_t mtime) : mem_(memory), mtime_(mtime)
{
	if (!memory)
		throw std::runtime_error("tar writer: no output buffer");
}

TarWriter::~TarWriter()
{
	// writer/mod.rs:84-96: best-effort finalisation on drop
	try {
		finish();
	} catch (...) {
	}
	if (file_ && own_)
		std::fclose(file_);
}

void TarWriter::put(const void *p, size_t n)
{
	if (n == 0)
		return;
	if (mem_) {
		const uint8_t *b = (const uint8_t *)p;
		mem_->insert(mem_->end(), b, b + n);
	} else if (std::fwrite(p, 1, n, file_) != n) {
		throw std::runtime_error(std::string("tar writer: write failed: ") + std::strerror(errno));
	}
}

void TarWriter::header(const std::string &path, uint64_t size, uint64_t mode, uint8_t typeflag)
{
	uint8_t h[512];
	std::memset(h, 0, sizeof h);
	if (path.size() > 100) // tar.rs:160-171: never truncate an entry's name
		throw std::runtime_error("tar header field overflow: \"" + path + "\" is " + std::to_string(path.size()) + " bytes, max 100");
	std::memcpy(h, path.data(), path.size());
	put_octal(h + 100, 8, mode);
	put_octal(h + 108, 8, 0); // uid
	put_octal(h + 116, 8, 0); // gid
	put_octal(h + 124, 12, size);
	const uint64_t t = mtime_ >= 0 ? (uint64_t)mtime_ : (uint64_t)std::time(nullptr);
	put_octal(h + 136, 12, t);
	h[156] = typeflag;
	std::memcpy(h + 257, "ustar\0", 6);
	std::memcpy(h + 263, "00", 2);
	std::memset(h + 148, ' ', 8); // the checksum is taken with its own field blank
	uint32_t sum = 0;
	for (uint8_t b : h)
		sum += b;
	put_octal(h + 148, 8, sum);
	put(h, sizeof h);
}

void TarWriter::write_file(const std::string &path, const uint8_t *data, size_t len)
{
	static const uint8_t zeros[512] = {0};
	header(path, len, 0644, '0');
	put(data, len);
	if (len % 512)
		put(zeros, 512 - len % 512);
}

void TarWriter::write_file(const std::string &path, const std::vector<uint8_t> &data)
{
	write_file(path, data.data(), data.size());
}

void TarWriter::write_gather(const std::string &path, const Piece *pieces, size_t n)
{
	static const uint8_t zeros[512] = {0};
	size_t len = 0;
	for (size_t i = 0; i < n; i++)
		len += pieces[i].second;
	header(path, len, 0644, '0');
	for (size_t i = 0; i < n; i++)
		put(pieces[i].first, pieces[i].second);
	if (len % 512)
		put(zeros, 512 - len % 512);
}

void TarWriter::write_directory(const std::string &path)
{
	if (path.empty() || path.back() != '/') // tar.rs:124
		throw std::runtime_error("dirname must end with a slash");
	header(path, 0, 0755, '5');
}

void TarWriter::finish()
{
	if (finished_) // writer/mod.rs:71-77: idempotent, no second trailer
		return;
	finished_ = true;
	static const uint8_t zeros[1024] = {0};
	put(zeros, sizeof zeros);
	if (file_ && std::fflush(file_) != 0)
		throw std::runtime_error(std::string("tar writer: flush failed: ") + std::strerror(errno));
}

// ---- directory sink ---------------------------------------------------------------------
namespace {

std::string join(const std::string &folder, const std::string &name)
{
	if (folder.empty())
		return name;
	if (!name.empty() && name[0] == '/') // PathBuf::join: an absolute component replaces the base
		return name;
	return folder.back() == '/' ? folder + name : folder + "/" + name;
}

void mkdir_all(const std::string &path)
{
	std::string cur;
	for (size_t i = 0; i <= path.size(); i++) {
		if (i == path.size() || path[i] == '/') {
			if (!cur.empty() && cur != "/") {
				if (::mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST)
					throw std::runtime_error("creating directory \"" + path + "\": " + std::strerror(errno));
				struct stat st;
				if (::stat(cur.c_str(), &st) != 0 || !S_ISDIR(st.st_mode))
					throw std::runtime_error("creating directory \"" + path + "\": not a directory: " + cur);
			}
		}
		if (i < path.size())
			cur.push_back(path[i]);
	}
}

} // namespace

void FileWriter::write_directory(const std::string &path) { mkdir_all(join(folder_, path)); }

void FileWriter::write_file(const std::string &path, const uint8_t *data, size_t len)
{
	const std::string full = join(folder_, path);
	std::FILE *f = std::fopen(full.c_str(), "wb");
	if (!f)
		throw std::runtime_error("writing \"" + full + "\": " + std::strerror(errno));
	const bool ok = len == 0 || std::fwrite(data, 1, len, f) == len;
	if (std::fclose(f) != 0 || !ok)
		throw std::runtime_error("writing \"" + full + "\": " + std::strerror(errno));
}

void FileWriter::write_file(const std::string &path, const std::vector<uint8_t> &data)
{
	write_file(path, data.data(), data.size());
}

void FileWriter::write_gather(const std::string &path, const Piece *pieces, size_t n)
{
	const std::string full = join(folder_, path);
	std::FILE *f = std::fopen(full.c_str(), "wb");
	if (!f)
		throw std::runtime_error("writing \"" + full + "\": " + std::strerror(errno));
	bool ok = true;
	for (size_t i = 0; i < n && ok; i++)
		ok = pieces[i].second == 0 || std::fwrite(pieces[i].first, 1, pieces[i].second, f) == pieces[i].second;
	if (std::fclose(f) != 0 || !ok)
		throw std::runtime_error("writing \"" + full + "\": " + std::strerror(errno));
}

} // namespace vg
