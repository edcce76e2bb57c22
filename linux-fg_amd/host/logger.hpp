// Logger -- the logging CONVENTION of the reference (src/logger.hpp:33-41,70-73), not its code: the four
// LOG_* macros, a process-wide Logger::Get(), and LOG_ERROR latching a last-error string that callers read with
// HasError() / GetLastError() / ClearError() after a method has returned false.
//
// Written for a headless frame pump rather than a desktop tool:
//   * lines go to stderr in one write(2)-sized fwrite, so a pixel stream on stdout (lfg_host --output-raw -) stays clean;
//   * the level filter is an atomic read taken before any formatting or locking -- the reference formats a
//     timestamp and takes a mutex for each of its ~9 per-frame INFO lines (src/scaler.cpp:262,360,465-477);
//   * time stamps are wall-clock HH:MM:SS.mmm (strftime on localtime_r), the latched error text carries none.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <ctime>
#include <mutex>
#include <sstream>
#include <string>
#include <utility>

class Logger {
public:
    enum class Level : int { DEBUG = 0, INFO = 1, WARNING = 2, ERROR = 3 };

    static Logger& Get() {
        static Logger theLogger;
        return theLogger;
    }

    template <typename... Parts>
    void Log(Level level, Parts&&... parts) {
        const bool isError = level == Level::ERROR;
        if (!isError && static_cast<int>(level) < m_threshold.load(std::memory_order_relaxed)) return;
        std::ostringstream text;
        Append(text, std::forward<Parts>(parts)...);
        Emit(level, text.str(), isError);
    }

    bool HasError() const {
        std::lock_guard<std::mutex> guard(m_lock);
        return m_errorLatched;
    }
    std::string GetLastError() const {
        std::lock_guard<std::mutex> guard(m_lock);
        return m_errorText;
    }
    void ClearError() {
        std::lock_guard<std::mutex> guard(m_lock);
        m_errorLatched = false;
        m_errorText.clear();
    }
    // Not in the reference: silence everything below `level` (errors are always recorded).
    void SetMinLevel(Level level) { m_threshold.store(static_cast<int>(level), std::memory_order_relaxed); }

private:
    Logger() = default;
    Logger(const Logger&) = delete;
    Logger& operator=(const Logger&) = delete;

    static void Append(std::ostringstream&) {}
    template <typename First, typename... Rest>
    static void Append(std::ostringstream& out, First&& first, Rest&&... rest) {
        out << std::forward<First>(first);
        Append(out, std::forward<Rest>(rest)...);
    }

    static const char* Name(Level level) {
        static const char* const kNames[] = {"DEBUG", "INFO", "WARNING", "ERROR"};
        const int i = static_cast<int>(level);
        return i >= 0 && i < 4 ? kNames[i] : "?";
    }

    void Emit(Level level, const std::string& text, bool latch) {
        using namespace std::chrono;
        const auto now = system_clock::now();
        const std::time_t secs = system_clock::to_time_t(now);
        const int millis = static_cast<int>(duration_cast<milliseconds>(now.time_since_epoch()).count() % 1000);
        std::tm local{};
        localtime_r(&secs, &local);
        char clock[16];
        std::strftime(clock, sizeof clock, "%H:%M:%S", &local);
        char head[48];
        const int headLen = std::snprintf(head, sizeof head, "%s.%03d %-7s ", clock, millis, Name(level));
        std::string line;
        line.reserve(static_cast<size_t>(headLen) + text.size() + 1);
        line.append(head, static_cast<size_t>(headLen)).append(text).push_back('\n');

        std::lock_guard<std::mutex> guard(m_lock);
        std::fwrite(line.data(), 1, line.size(), stderr);
        if (latch) {
            m_errorLatched = true;
            m_errorText = text;
        }
    }

    mutable std::mutex m_lock;
    std::atomic<int> m_threshold{static_cast<int>(Level::INFO)};
    bool m_errorLatched = false;
    std::string m_errorText;
};

#define LOG_DEBUG(...) Logger::Get().Log(Logger::Level::DEBUG, __VA_ARGS__)
#define LOG_INFO(...) Logger::Get().Log(Logger::Level::INFO, __VA_ARGS__)
#define LOG_WARN(...) Logger::Get().Log(Logger::Level::WARNING, __VA_ARGS__)
#define LOG_ERROR(...) Logger::Get().Log(Logger::Level::ERROR, __VA_ARGS__)
